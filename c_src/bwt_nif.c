/* Erlang NIF shim over libbwtgpu (include/bwt_gpu.h).  The ONLY file that includes erl_nif.h.
 *
 * NOT COMPILED in the authoring container (no Erlang/OTP there: no erl_nif.h); kept as the
 * reference-side binding a maintainer adds.  Build where OTP is installed:
 *     cc -O2 -fPIC -shared -I$ERL_ROOT/usr/include -I../include bwt_nif.c -L../bwt_amd -lbwtgpu \
 *        -Wl,-rpath,'$ORIGIN' -o priv/bwt_nif.so
 *
 * Exposed to Erlang (module bwt_nif, see erl/bwt_nif.erl):
 *   constants() -> {Tolerance,MinLen,MaxRange,MaxLen,Match,Mismatch,Gap,GapExt,Unknown,RefExt}
 *   index_open(FmPath, RefPath, Device) -> {ok, Index} | {error, {Code, Text}}
 *   index_meta(Index) -> {Pc,Pg,Pt,Last,Shift}
 *   align_batch(Index, [{Name, Seq}]) ->
 *        [{Name, nothing | {error, Reason} | [{{Score,Cigar}, Pos, RefSeq}]}]      (reference list order)
 * align_batch (bwt_batch_set_reads + bwt_batch_run + bwt_batch_fetch_hits) replaces the two per-read calls of worker_bwt.erl (sga:sga/6 at :76, sw:sw/2 at :92)
 * and runs on a dirty IO scheduler (a batch of 200 reads x 10 lists takes well over 1 ms).
 */
#include <erl_nif.h>
#include <stdlib.h>
#include <string.h>
#include "bwt_gpu.h"

static ErlNifResourceType *INDEX_RT;
typedef struct { bwt_index *ix; bwt_batch *batch; } index_res;   /* one resident batch per index handle */

static void index_dtor(ErlNifEnv *env, void *obj) {
    (void)env;
    bwt_batch_destroy(((index_res *)obj)->batch);
    bwt_index_close(((index_res *)obj)->ix);
}

static int load(ErlNifEnv *env, void **priv, ERL_NIF_TERM info) {
    (void)priv; (void)info;
    INDEX_RT = enif_open_resource_type(env, NULL, "bwt_index", index_dtor, ERL_NIF_RT_CREATE, NULL);
    return INDEX_RT ? 0 : 1;
}

static ERL_NIF_TERM mk_error(ErlNifEnv *env, int code) {
    return enif_make_tuple2(env, enif_make_atom(env, "error"),
                            enif_make_tuple2(env, enif_make_int(env, code),
                                             enif_make_string(env, bwt_last_error(), ERL_NIF_LATIN1)));
}

static ERL_NIF_TERM nif_constants(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc; (void)argv;
    bwt_constants_t c;
    bwt_constants(&c);
    return enif_make_tuple(env, 10, enif_make_int(env, c.tolerance), enif_make_int(env, c.min_len),
                           enif_make_int(env, c.max_range), enif_make_int(env, c.max_len),
                           enif_make_int(env, c.match), enif_make_int(env, c.mismatch),
                           enif_make_int(env, c.gap_penalty), enif_make_int(env, c.gap_ext_penalty),
                           enif_make_double(env, c.unknown), enif_make_int(env, c.ref_extension_len));
}

static ERL_NIF_TERM nif_index_open(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
    char fm[4096], ref[4096];
    int dev;
    (void)argc;
    if (enif_get_string(env, argv[0], fm, sizeof fm, ERL_NIF_LATIN1) <= 0 ||
        enif_get_string(env, argv[1], ref, sizeof ref, ERL_NIF_LATIN1) <= 0 || !enif_get_int(env, argv[2], &dev))
        return enif_make_badarg(env);
    bwt_index *ix = NULL;
    int rc = bwt_index_open(fm, ref, dev, &ix);
    if (rc) return mk_error(env, rc);
    index_res *r = enif_alloc_resource(INDEX_RT, sizeof *r);
    r->ix = ix; r->batch = NULL;
    if ((rc = bwt_batch_create(ix, &r->batch))) { enif_release_resource(r); return mk_error(env, rc); }
    ERL_NIF_TERM t = enif_make_resource(env, r);
    enif_release_resource(r);
    return enif_make_tuple2(env, enif_make_atom(env, "ok"), t);
}

static ERL_NIF_TERM nif_index_meta(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
    index_res *r;
    uint32_t m[5];
    (void)argc;
    if (!enif_get_resource(env, argv[0], INDEX_RT, (void **)&r)) return enif_make_badarg(env);
    if (bwt_index_meta(r->ix, m)) return mk_error(env, BWT_EINVAL);
    return enif_make_tuple5(env, enif_make_uint(env, m[0]), enif_make_uint(env, m[1]), enif_make_uint(env, m[2]),
                            enif_make_uint(env, m[3]), enif_make_int(env, (int32_t)m[4]));
}

static ERL_NIF_TERM nif_align_batch(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
    index_res *r;
    unsigned n;
    (void)argc;
    if (!enif_get_resource(env, argv[0], INDEX_RT, (void **)&r) || !enif_get_list_length(env, argv[1], &n))
        return enif_make_badarg(env);
    /* copy names and sequences (lists of bytes or binaries: iolists) into one staging buffer */
    ERL_NIF_TERM *names = malloc(sizeof(ERL_NIF_TERM) * (n ? n : 1));
    uint64_t *off = malloc(sizeof(uint64_t) * (n + 1));
    size_t cap = 256 * (size_t)(n ? n : 1), used = 0;
    uint8_t *bases = malloc(cap);
    ERL_NIF_TERM head, tail = argv[1];
    unsigned i = 0;
    off[0] = 0;
    while (enif_get_list_cell(env, tail, &head, &tail)) {
        const ERL_NIF_TERM *tp;
        int ar;
        ErlNifBinary bin;
        if (!enif_get_tuple(env, head, &ar, &tp) || ar != 2 || !enif_inspect_iolist_as_binary(env, tp[1], &bin)) {
            free(names); free(off); free(bases);
            return enif_make_badarg(env);
        }
        if (used + bin.size > cap) { cap = (used + bin.size) * 2; bases = realloc(bases, cap); }
        memcpy(bases + used, bin.data, bin.size);
        used += bin.size;
        names[i] = tp[0];
        off[++i] = used;
    }
    int32_t *rstat = malloc(sizeof(int32_t) * (n ? n : 1));
    uint64_t *hoff = malloc(sizeof(uint64_t) * (n + 1));
    uint64_t need[2] = {0, 0};
    /* resident-batch interface: upload, run the three stages once, then size and fetch the hit tables
     * (bwt_align_batch's two-call pattern would run the pipeline twice).  One call at a time per handle. */
    int rc = bwt_batch_set_reads(r->batch, bases, off, n);
    if (!rc) rc = bwt_batch_run(r->batch, BWT_STAGE_EXTEND, NULL);
    int32_t *score = NULL, *wlen = NULL; int64_t *pos = NULL, *woff = NULL; uint64_t *coff = NULL; char *cig = NULL;
    if (!rc) {
        rc = bwt_batch_fetch_hits(r->batch, rstat, hoff, 0, 0, 0, 0, 0, 0, 0, 0, need);
        if (rc == BWT_ENOSPC || rc == 0) {
            score = malloc(4 * (need[0] + 1)); wlen = malloc(4 * (need[0] + 1));
            pos = malloc(8 * (need[0] + 1)); woff = malloc(8 * (need[0] + 1));
            coff = malloc(8 * (need[0] + 2)); cig = malloc(need[1] + 1);
            rc = bwt_batch_fetch_hits(r->batch, rstat, hoff, score, pos, woff, wlen, coff, cig, need[0], need[1], need);
        }
    }
    /* RefSeq of every hit in one device gather */
    uint8_t *wins = NULL; uint64_t *woffs = NULL;
    if (!rc && need[0]) {
        uint64_t tot = 0;
        woffs = malloc(8 * (need[0] + 1));
        rc = bwt_ref1_gather(r->ix, woff, wlen, need[0], NULL, 0, woffs, &tot);
        if (rc == BWT_ENOSPC) { wins = malloc(tot + 1); rc = bwt_ref1_gather(r->ix, woff, wlen, need[0], wins, tot, woffs, &tot); }
    }
    ERL_NIF_TERM out = enif_make_list(env, 0);
    if (rc) out = mk_error(env, rc);
    else {
        for (i = n; i-- > 0;) {                       /* build the list back to front */
            ERL_NIF_TERM v;
            if (rstat[i] == 1) v = enif_make_atom(env, "nothing");
            else if (rstat[i] == -1) v = enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "bad_base"));
            else if (rstat[i] == -2) v = enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "window_out_of_range"));
            else {
                v = enif_make_list(env, 0);
                for (uint64_t h = hoff[i + 1]; h-- > hoff[i];) {
                    const uint8_t *win = wins + woffs[h];
                    ERL_NIF_TERM sc = enif_make_tuple2(env, enif_make_int(env, score[h]),
                        enif_make_string_len(env, cig + coff[h], (size_t)(coff[h + 1] - coff[h]), ERL_NIF_LATIN1));
                    ERL_NIF_TERM t = enif_make_tuple3(env, sc, enif_make_int64(env, pos[h]),
                        enif_make_string_len(env, (char *)win, (size_t)wlen[h], ERL_NIF_LATIN1));
                    v = enif_make_list_cell(env, t, v);
                }
            }
            out = enif_make_list_cell(env, enif_make_tuple2(env, names[i], v), out);
        }
    }
    free(names); free(off); free(bases); free(rstat); free(hoff);
    free(score); free(wlen); free(pos); free(woff); free(coff); free(cig); free(wins); free(woffs);
    return out;
}

static ErlNifFunc funcs[] = {
    {"constants", 0, nif_constants, 0},
    {"index_open", 3, nif_index_open, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"index_meta", 1, nif_index_meta, 0},
    {"align_batch", 2, nif_align_batch, ERL_NIF_DIRTY_JOB_IO_BOUND},
};
ERL_NIF_INIT(bwt_nif, funcs, load, NULL, NULL, NULL)

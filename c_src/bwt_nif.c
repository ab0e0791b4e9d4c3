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
 *        | {error, enomem} | {error, {Code, Text}}
 * Thread safety: calls on the same Index term are serialised by a mutex in the resource (the resident batch is not
 * re-entrant); calls on different Index terms, and everything below the batch in libbwtgpu, run concurrently.
 * align_batch (bwt_batch_set_reads + bwt_batch_run + bwt_batch_fetch_hits) replaces the two per-read calls of worker_bwt.erl (sga:sga/6 at :76, sw:sw/2 at :92)
 * and runs on a dirty IO scheduler (a batch of 200 reads x 10 lists takes well over 1 ms).
 */
#include <erl_nif.h>
#include <stdlib.h>
#include <string.h>
#include "bwt_gpu.h"

static ErlNifResourceType *INDEX_RT;
/* One resident batch per index handle (its device buffers are reused from call to call).  A bwt_batch is not
 * re-entrant (include/bwt_gpu.h, threading), and BEAM may run align_batch for the same Index term on several
 * dirty schedulers at once: `lock` serialises them.  Different Index terms run concurrently. */
typedef struct { bwt_index *ix; bwt_batch *batch; ErlNifMutex *lock; } index_res;

static void index_dtor(ErlNifEnv *env, void *obj) {
    index_res *r = (index_res *)obj;
    (void)env;
    if (r->batch) bwt_batch_destroy(r->batch);
    if (r->ix) bwt_index_close(r->ix);
    if (r->lock) enif_mutex_destroy(r->lock);
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
    if (!r) { bwt_index_close(ix); return enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "enomem")); }
    r->ix = ix; r->batch = NULL;
    r->lock = enif_mutex_create("bwt_index_batch");
    if (!r->lock) { enif_release_resource(r); return enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "enomem")); }
    if ((rc = bwt_batch_create(ix, &r->batch))) { ERL_NIF_TERM e = mk_error(env, rc); enif_release_resource(r); return e; }
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

/* every buffer of one align_batch call; freed in one place */
typedef struct {
    ERL_NIF_TERM *names; uint64_t *off; uint8_t *bases; int32_t *rstat; uint64_t *hoff;
    int32_t *score, *wlen; int64_t *pos, *woff; uint64_t *coff; char *cig; uint8_t *wins; uint64_t *woffs;
    uint32_t *codes; uint64_t *exc;                                     /* the packed form of the reads (what crosses PCIe) */
    int8_t *rstat8; bwt_hit_rec *hits; bwt_hit_extra *extra; char *xcig;   /* the compact hit table (what comes back) */
} call_bufs;
static void bufs_free(call_bufs *b) {
    free(b->names); free(b->off); free(b->bases); free(b->rstat); free(b->hoff); free(b->score); free(b->wlen);
    free(b->pos); free(b->woff); free(b->coff); free(b->cig); free(b->wins); free(b->woffs);
    free(b->codes); free(b->exc); free(b->rstat8); free(b->hits); free(b->extra); free(b->xcig);
}
static ERL_NIF_TERM enomem(ErlNifEnv *env, call_bufs *b) {
    bufs_free(b);
    return enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "enomem"));
}

static ERL_NIF_TERM nif_align_batch(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
    index_res *r;
    unsigned n;
    call_bufs B;
    (void)argc;
    memset(&B, 0, sizeof B);
    if (!enif_get_resource(env, argv[0], INDEX_RT, (void **)&r) || !enif_get_list_length(env, argv[1], &n))
        return enif_make_badarg(env);
    /* copy names and sequences (lists of bytes or binaries: iolists) into one staging buffer */
    size_t cap = 256 * (size_t)(n ? n : 1), used = 0;
    B.names = malloc(sizeof(ERL_NIF_TERM) * (n ? n : 1));
    B.off = malloc(sizeof(uint64_t) * ((size_t)n + 1));
    B.bases = malloc(cap);
    B.rstat = malloc(sizeof(int32_t) * (n ? n : 1));
    B.hoff = malloc(sizeof(uint64_t) * ((size_t)n + 1));
    if (!B.names || !B.off || !B.bases || !B.rstat || !B.hoff) return enomem(env, &B);
    ERL_NIF_TERM head, tail = argv[1];
    unsigned i = 0;
    B.off[0] = 0;
    while (enif_get_list_cell(env, tail, &head, &tail)) {
        const ERL_NIF_TERM *tp;
        int ar;
        ErlNifBinary bin;
        if (i >= n || !enif_get_tuple(env, head, &ar, &tp) || ar != 2 || !enif_inspect_iolist_as_binary(env, tp[1], &bin)) {
            bufs_free(&B);
            return enif_make_badarg(env);
        }
        if (used + bin.size > cap) {
            size_t ncap = (used + bin.size) * 2;
            uint8_t *nb = realloc(B.bases, ncap);
            if (!nb) return enomem(env, &B);
            B.bases = nb; cap = ncap;
        }
        memcpy(B.bases + used, bin.data, bin.size);
        used += bin.size;
        B.names[i] = tp[0];
        B.off[++i] = used;
    }
    if (i != n) { bufs_free(&B); return enif_make_badarg(env); }       /* improper list */
    uint64_t need[2] = {0, 0};
    /* The strings become the packed wire format on the host (2 bits per base + a list of the bases outside ACGT), go up, run the
     * three stages once, and the compact hit table comes back (16 bytes per hit; window and CIGAR only for the hits that are not
     * "<Lq>M" on a window of Lq bases) -- expanded here into the per-hit arrays the terms are built from.
     * The handle's batch is used under its lock. */
    int nomem = 0;
    uint64_t n_exc = 0, xneed[3] = {0, 0, 0}, exc_cap = used / 64 + 64;
    B.codes = malloc(4 * (size_t)bwt_packed_words(used));
    B.exc = malloc(8 * (size_t)exc_cap);
    B.rstat8 = malloc((size_t)n + 1);
    if (!B.codes || !B.exc || !B.rstat8) return enomem(env, &B);
    int rc = bwt_pack_reads(B.bases, used, B.codes, B.exc, exc_cap, &n_exc, 1);
    if (rc == BWT_ENOSPC) {                            /* many N: the count is known now */
        free(B.exc);
        exc_cap = n_exc;
        B.exc = malloc(8 * (size_t)exc_cap + 8);
        if (!B.exc) return enomem(env, &B);
        rc = bwt_pack_reads(B.bases, used, B.codes, B.exc, exc_cap, &n_exc, 1);
    }
    if (rc) { ERL_NIF_TERM e = mk_error(env, rc); bufs_free(&B); return e; }
    uint32_t shift_meta[5] = {0, 0, 0, 0, 0};
    (void)bwt_index_meta(r->ix, shift_meta);
    enif_mutex_lock(r->lock);
    rc = bwt_batch_set_reads_packed(r->batch, B.codes, used, n, 0, B.off, B.exc, n_exc);
    if (!rc) rc = bwt_batch_run(r->batch, BWT_STAGE_EXTEND, NULL);
    if (!rc) {
        rc = bwt_batch_fetch_hits_compact(r->batch, 0, 0, 0, 0, 0, 0, 0, xneed);
        if (rc == BWT_ENOSPC || rc == 0) {
            B.hits = malloc(sizeof(bwt_hit_rec) * (xneed[0] + 1)); B.extra = malloc(sizeof(bwt_hit_extra) * (xneed[1] + 1));
            B.xcig = malloc(xneed[2] + 1);
            if (!B.hits || !B.extra || !B.xcig) nomem = 1;
            else rc = bwt_batch_fetch_hits_compact(r->batch, B.rstat8, B.hits, B.extra, B.xcig, xneed[0], xneed[1], xneed[2], xneed);
        }
    }
    enif_mutex_unlock(r->lock);
    if (nomem) return enomem(env, &B);
    if (!rc) {
        for (i = 0; i < n; i++) B.rstat[i] = B.rstat8[i];
        need[0] = xneed[0];
        B.score = malloc(4 * (need[0] + 1)); B.wlen = malloc(4 * (need[0] + 1));
        B.pos = malloc(8 * (need[0] + 1)); B.woff = malloc(8 * (need[0] + 1));
        B.coff = malloc(8 * (need[0] + 2));
        if (!B.score || !B.wlen || !B.pos || !B.woff || !B.coff) return enomem(env, &B);
        rc = bwt_hits_expand(n, 0, B.off, (int32_t)shift_meta[4], B.hits, xneed[0], B.extra, xneed[1], B.xcig, xneed[2], B.hoff, B.score, B.pos,
                             B.woff, B.wlen, B.coff, NULL, 0, &need[1]);
        if (!rc) {
            B.cig = malloc(need[1] + 1);
            if (!B.cig) return enomem(env, &B);
            rc = bwt_hits_expand(n, 0, B.off, (int32_t)shift_meta[4], B.hits, xneed[0], B.extra, xneed[1], B.xcig, xneed[2], B.hoff, B.score,
                                 B.pos, B.woff, B.wlen, B.coff, B.cig, need[1], &need[1]);
        }
    }
    /* RefSeq of every hit in one device gather (index-level call: thread-safe, no batch involved) */
    if (!rc && need[0]) {
        uint64_t tot = 0;
        B.woffs = malloc(8 * (need[0] + 1));
        if (!B.woffs) return enomem(env, &B);
        rc = bwt_ref1_gather(r->ix, B.woff, B.wlen, need[0], NULL, 0, B.woffs, &tot);
        if (rc == BWT_ENOSPC) {
            B.wins = malloc(tot + 1);
            if (!B.wins) return enomem(env, &B);
            rc = bwt_ref1_gather(r->ix, B.woff, B.wlen, need[0], B.wins, tot, B.woffs, &tot);
        }
    }
    ERL_NIF_TERM out = enif_make_list(env, 0);
    if (rc) out = mk_error(env, rc);
    else {
        for (i = n; i-- > 0;) {                       /* build the list back to front */
            ERL_NIF_TERM v;
            if (B.rstat[i] == 1) v = enif_make_atom(env, "nothing");
            else if (B.rstat[i] == -1) v = enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "bad_base"));
            else if (B.rstat[i] == -2) v = enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "window_out_of_range"));
            else {
                v = enif_make_list(env, 0);
                for (uint64_t h = B.hoff[i + 1]; h-- > B.hoff[i];) {
                    const uint8_t *win = B.wins + B.woffs[h];
                    ERL_NIF_TERM sc = enif_make_tuple2(env, enif_make_int(env, B.score[h]),
                        enif_make_string_len(env, B.cig + B.coff[h], (size_t)(B.coff[h + 1] - B.coff[h]), ERL_NIF_LATIN1));
                    ERL_NIF_TERM t = enif_make_tuple3(env, sc, enif_make_int64(env, B.pos[h]),
                        enif_make_string_len(env, (const char *)win, (size_t)B.wlen[h], ERL_NIF_LATIN1));
                    v = enif_make_list_cell(env, t, v);
                }
            }
            out = enif_make_list_cell(env, enif_make_tuple2(env, B.names[i], v), out);
        }
    }
    bufs_free(&B);
    return out;
}

static ErlNifFunc funcs[] = {
    {"constants", 0, nif_constants, 0},
    {"index_open", 3, nif_index_open, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"index_meta", 1, nif_index_meta, 0},
    {"align_batch", 2, nif_align_batch, ERL_NIF_DIRTY_JOB_IO_BOUND},
};
ERL_NIF_INIT(bwt_nif, funcs, load, NULL, NULL, NULL)

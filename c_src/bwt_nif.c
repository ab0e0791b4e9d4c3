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
 *   align_workload(Index, [[{Name, Seq}]]) -> [what align_batch returns for each list]: a whole {workload, Workload} message in one call
 * align_batch (one bwt_small_align call: 0.15 ms for a QseqList of 200 reads, 0.23 ms for a workload of 2000; larger batches go
 * through the general path inside the same call) replaces the two per-read calls of worker_bwt.erl (sga:sga/6 at :76, sw:sw/2 at
 * :92) and runs on a dirty IO scheduler.
 */
#include <erl_nif.h>
#include <stdlib.h>
#include <string.h>
#include "bwt_gpu.h"

static ErlNifResourceType *INDEX_RT;
/* One small-batch handle per index term (its device and pinned buffers are reused from call to call).  A bwt_small is not
 * re-entrant (include/bwt_gpu.h, threading), and BEAM may run align_batch for the same Index term on several
 * dirty schedulers at once: `lock` serialises them.  Different Index terms run concurrently. */
typedef struct { bwt_index *ix; bwt_small *small; ErlNifMutex *lock; } index_res;

static void index_dtor(ErlNifEnv *env, void *obj) {
    index_res *r = (index_res *)obj;
    (void)env;
    if (r->small) bwt_small_destroy(r->small);
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
    r->ix = ix; r->small = NULL;
    r->lock = enif_mutex_create("bwt_index_batch");
    if (!r->lock) { enif_release_resource(r); return enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "enomem")); }
    if ((rc = bwt_small_create(ix, &r->small))) { ERL_NIF_TERM e = mk_error(env, rc); enif_release_resource(r); return e; }
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

/* every buffer of one align call; freed in one place */
typedef struct {
    ERL_NIF_TERM *names; uint64_t *off; uint8_t *bases; int32_t *rstat; uint64_t *hoff;
    int32_t *score, *wlen; int64_t *pos, *woff; uint64_t *coff; char *cig; uint8_t *wins; uint64_t *woffs;
    unsigned *list_len;
} call_bufs;
static void bufs_free(call_bufs *b) {
    free(b->names); free(b->off); free(b->bases); free(b->rstat); free(b->hoff); free(b->score); free(b->wlen);
    free(b->pos); free(b->woff); free(b->coff); free(b->cig); free(b->wins); free(b->woffs); free(b->list_len);
}
static ERL_NIF_TERM enomem(ErlNifEnv *env, call_bufs *b) {
    bufs_free(b);
    return enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "enomem"));
}

/* appends the {Name, Seq} items of one QseqList to the staging buffers; 0 = badarg */
static int stage_list(ErlNifEnv *env, ERL_NIF_TERM list, call_bufs *B, unsigned *i, unsigned n, size_t *used, size_t *cap, int *nomem) {
    ERL_NIF_TERM head, tail = list;
    while (enif_get_list_cell(env, tail, &head, &tail)) {
        const ERL_NIF_TERM *tp;
        int ar;
        ErlNifBinary bin;
        if (*i >= n || !enif_get_tuple(env, head, &ar, &tp) || ar != 2 || !enif_inspect_iolist_as_binary(env, tp[1], &bin)) return 0;
        if (*used + bin.size > *cap) {
            size_t ncap = (*used + bin.size) * 2;
            uint8_t *nb = realloc(B->bases, ncap);
            if (!nb) { *nomem = 1; return 0; }
            B->bases = nb; *cap = ncap;
        }
        memcpy(B->bases + *used, bin.data, bin.size);
        *used += bin.size;
        B->names[*i] = tp[0];
        B->off[++*i] = *used;
    }
    return 1;
}

/* The reads of nlists QseqLists (lens[k] of them each, n in all, staged in B) through ONE bwt_small_align call -- the small-batch
 * entry point takes a QseqList (200 reads, src/master.erl:59) or a whole workload message (ten of them) in 0.15-0.25 ms and runs
 * anything larger through the general path itself -- and the result terms, one list per QseqList:
 *     [{Name, nothing | {error, Reason} | [{{Score,Cigar}, Pos, RefSeq}]}]          (reference list order)
 * out[k] receives list k's term.  Returns 0, or an error term in *err. */
static int align_staged(ErlNifEnv *env, index_res *r, call_bufs *B, unsigned n, const unsigned *lens, unsigned nlists, ERL_NIF_TERM *out,
                        ERL_NIF_TERM *err) {
    uint64_t need[2] = {0, 0};
    uint32_t meta[5] = {0, 0, 0, 0, 0};
    unsigned i;
    int rc, nomem = 0;
    (void)bwt_index_meta(r->ix, meta);
    B->rstat = malloc(sizeof(int32_t) * (n ? n : 1));
    B->hoff = malloc(sizeof(uint64_t) * ((size_t)n + 1));
    if (!B->rstat || !B->hoff) { *err = enomem(env, B); return 1; }
    /* the handle's small-batch state is used under its lock; the result pointers are the handle's (valid until its next call):
     * the compact hit table is expanded into this call's own arrays before the lock is released */
    enif_mutex_lock(r->lock);
    bwt_small_result sr;
    rc = bwt_small_align(r->small, B->bases, B->off, n, &sr);
    if (!rc) {
        for (i = 0; i < n; i++) B->rstat[i] = sr.read_status[i];
        need[0] = sr.n_hits;
        B->score = malloc(4 * (need[0] + 1)); B->wlen = malloc(4 * (need[0] + 1));
        B->pos = malloc(8 * (need[0] + 1)); B->woff = malloc(8 * (need[0] + 1));
        B->coff = malloc(8 * (need[0] + 2));
        if (!B->score || !B->wlen || !B->pos || !B->woff || !B->coff) nomem = 1;
        else {
            rc = bwt_hits_expand(n, 0, B->off, (int32_t)meta[4], sr.hits, sr.n_hits, sr.extra, sr.n_extra, sr.cigar, sr.cigar_bytes, B->hoff, B->score,
                                 B->pos, B->woff, B->wlen, B->coff, NULL, 0, &need[1]);
            if (!rc) {
                B->cig = malloc(need[1] + 1);
                if (!B->cig) nomem = 1;
                else rc = bwt_hits_expand(n, 0, B->off, (int32_t)meta[4], sr.hits, sr.n_hits, sr.extra, sr.n_extra, sr.cigar, sr.cigar_bytes, B->hoff,
                                          B->score, B->pos, B->woff, B->wlen, B->coff, B->cig, need[1], &need[1]);
            }
        }
    }
    enif_mutex_unlock(r->lock);
    if (nomem) { *err = enomem(env, B); return 1; }
    /* RefSeq of every hit in one device gather (index-level call: thread-safe, no batch involved) */
    if (!rc && need[0]) {
        uint64_t tot = 0;
        B->woffs = malloc(8 * (need[0] + 1));
        if (!B->woffs) { *err = enomem(env, B); return 1; }
        rc = bwt_ref1_gather(r->ix, B->woff, B->wlen, need[0], NULL, 0, B->woffs, &tot);
        if (rc == BWT_ENOSPC) {
            B->wins = malloc(tot + 1);
            if (!B->wins) { *err = enomem(env, B); return 1; }
            rc = bwt_ref1_gather(r->ix, B->woff, B->wlen, need[0], B->wins, tot, B->woffs, &tot);
        }
    }
    if (rc) { *err = mk_error(env, rc); bufs_free(B); return 1; }
    unsigned base = n;
    for (unsigned k = nlists; k-- > 0;) {             /* every list back to front */
        ERL_NIF_TERM lst = enif_make_list(env, 0);
        base -= lens[k];
        for (i = base + lens[k]; i-- > base;) {
            ERL_NIF_TERM v;
            if (B->rstat[i] == 1) v = enif_make_atom(env, "nothing");
            else if (B->rstat[i] == -1) v = enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "bad_base"));
            else if (B->rstat[i] == -2) v = enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, "window_out_of_range"));
            else {
                v = enif_make_list(env, 0);
                for (uint64_t h = B->hoff[i + 1]; h-- > B->hoff[i];) {
                    const uint8_t *win = B->wins + B->woffs[h];
                    ERL_NIF_TERM sc = enif_make_tuple2(env, enif_make_int(env, B->score[h]),
                        enif_make_string_len(env, B->cig + B->coff[h], (size_t)(B->coff[h + 1] - B->coff[h]), ERL_NIF_LATIN1));
                    ERL_NIF_TERM t = enif_make_tuple3(env, sc, enif_make_int64(env, B->pos[h]),
                        enif_make_string_len(env, (const char *)win, (size_t)B->wlen[h], ERL_NIF_LATIN1));
                    v = enif_make_list_cell(env, t, v);
                }
            }
            lst = enif_make_list_cell(env, enif_make_tuple2(env, B->names[i], v), lst);
        }
        out[k] = lst;
    }
    bufs_free(B);
    return 0;
}

static ERL_NIF_TERM nif_align_batch(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
    index_res *r;
    unsigned n, i = 0;
    call_bufs B;
    int nomem = 0;
    (void)argc;
    memset(&B, 0, sizeof B);
    if (!enif_get_resource(env, argv[0], INDEX_RT, (void **)&r) || !enif_get_list_length(env, argv[1], &n))
        return enif_make_badarg(env);
    /* copy names and sequences (lists of bytes or binaries: iolists) into one staging buffer */
    size_t cap = 256 * (size_t)(n ? n : 1), used = 0;
    B.names = malloc(sizeof(ERL_NIF_TERM) * (n ? n : 1));
    B.off = malloc(sizeof(uint64_t) * ((size_t)n + 1));
    B.bases = malloc(cap);
    if (!B.names || !B.off || !B.bases) return enomem(env, &B);
    B.off[0] = 0;
    if (!stage_list(env, argv[1], &B, &i, n, &used, &cap, &nomem) || i != n) {       /* (i != n: improper list) */
        if (nomem) return enomem(env, &B);
        bufs_free(&B);
        return enif_make_badarg(env);
    }
    ERL_NIF_TERM out, err;
    if (align_staged(env, r, &B, n, &n, 1, &out, &err)) return err;
    return out;
}

/* align_workload(Index, [QseqList]) -> [ResultList]: a whole {workload, Workload} message (src/worker_bwt.erl:62-69) in one device
 * call; element k is what align_batch(Index, lists:nth(k, Workload)) returns. */
static ERL_NIF_TERM nif_align_workload(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
    index_res *r;
    unsigned nlists, n = 0, i = 0, k = 0;
    call_bufs B;
    int nomem = 0;
    (void)argc;
    memset(&B, 0, sizeof B);
    if (!enif_get_resource(env, argv[0], INDEX_RT, (void **)&r) || !enif_get_list_length(env, argv[1], &nlists))
        return enif_make_badarg(env);
    B.list_len = malloc(sizeof(unsigned) * (nlists ? nlists : 1));
    if (!B.list_len) return enomem(env, &B);
    ERL_NIF_TERM head, tail = argv[1];
    while (enif_get_list_cell(env, tail, &head, &tail)) {
        unsigned len;
        if (k >= nlists || !enif_get_list_length(env, head, &len)) { bufs_free(&B); return enif_make_badarg(env); }
        B.list_len[k++] = len;
        n += len;
    }
    if (k != nlists) { bufs_free(&B); return enif_make_badarg(env); }
    size_t cap = 256 * (size_t)(n ? n : 1), used = 0;
    B.names = malloc(sizeof(ERL_NIF_TERM) * (n ? n : 1));
    B.off = malloc(sizeof(uint64_t) * ((size_t)n + 1));
    B.bases = malloc(cap);
    ERL_NIF_TERM *outs = malloc(sizeof(ERL_NIF_TERM) * (nlists ? nlists : 1));
    if (!B.names || !B.off || !B.bases || !outs) { free(outs); return enomem(env, &B); }
    B.off[0] = 0;
    tail = argv[1];
    unsigned want = 0;
    k = 0;
    while (enif_get_list_cell(env, tail, &head, &tail)) {
        want += B.list_len[k++];
        if (!stage_list(env, head, &B, &i, n, &used, &cap, &nomem) || i != want) {
            free(outs);
            if (nomem) return enomem(env, &B);
            bufs_free(&B);
            return enif_make_badarg(env);
        }
    }
    unsigned *lens = B.list_len;
    B.list_len = NULL;                                  /* (align_staged frees B; the lengths are read while it builds the terms) */
    ERL_NIF_TERM err;
    if (align_staged(env, r, &B, n, lens, nlists, outs, &err)) { free(outs); free(lens); return err; }
    ERL_NIF_TERM res = enif_make_list(env, 0);
    for (k = nlists; k-- > 0;) res = enif_make_list_cell(env, outs[k], res);
    free(outs); free(lens);
    return res;
}

static ErlNifFunc funcs[] = {
    {"constants", 0, nif_constants, 0},
    {"index_open", 3, nif_index_open, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"index_meta", 1, nif_index_meta, 0},
    {"align_batch", 2, nif_align_batch, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"align_workload", 2, nif_align_workload, ERL_NIF_DIRTY_JOB_IO_BOUND},
};
ERL_NIF_INIT(bwt_nif, funcs, load, NULL, NULL, NULL)

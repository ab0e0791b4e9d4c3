/* libbwtgpu -- C ABI of the MI355X (gfx950) seed-search / clustering / Smith-Waterman path.
 *
 * This is the drop-in boundary for the hot path of cloudozer/BWT: every entry point replaces
 * one Erlang call site of the reference (paths relative to /root/reference) and is what a
 * NIF / port driver for that call site would bind (c_src/bwt_nif.c).
 *
 * Conventions
 *   - plain C, no exceptions cross the ABI; every function returns 0 or a negative BWT_E* code;
 *     bwt_last_error() returns a thread-local text for the last failure.
 *   - inputs are caller-owned host buffers, read-only for the duration of the call;
 *     outputs go into caller-provided host buffers with explicit capacities.  When a capacity
 *     is too small the call returns BWT_ENOSPC after filling the `needed` counts, and may be
 *     repeated with larger buffers (two-call pattern: pass capacity 0 first to size).
 *   - reads are concatenated ASCII bytes; `off` has n+1 entries, read i = bases[off[i] .. off[i+1]).
 *   - there is NO CPU fallback: every *_batch function runs HIP kernels on `device`.
 *   - threading: a bwt_index may be used by any number of host threads at once -- every host-buffer entry point
 *     (bwt_find_seeds_batch, bwt_sga_batch, bwt_align_batch, bwt_sw_batch, bwt_ref1_*) works on a private batch with
 *     its own HIP stream and does its own hipSetDevice, and holds the index's shared lock while kernels read it;
 *     bwt_index_set_ktable takes the exclusive lock (it waits for running batches).  A bwt_batch / bwt_pipe /
 *     bwt_fastq handle is NOT re-entrant: one call at a time per handle (use one handle per thread, or a lock
 *     around it as c_src/bwt_nif.c does).  bwt_index_close must not race with anything else on that index.
 *     bwt_last_error() is per thread.
 */
#ifndef BWT_GPU_H
#define BWT_GPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BWT_OK 0
#define BWT_EINVAL (-22)   /* bad argument */
#define BWT_ENOMEM (-12)   /* host or device allocation failed */
#define BWT_ENOSPC (-28)   /* an output capacity is too small; see `needed` */
#define BWT_EIO (-5)       /* file could not be read */
#define BWT_EFORMAT (-74)  /* .fm is not the term fm_index:make_index/4 writes */
#define BWT_EHIP (-71)     /* HIP runtime error (text in bwt_last_error) */
#define BWT_EOVERFLOW (-75)/* a batch produced more than 2^32-1 seeds/candidates: split it */

/* constants of include/bwt.hrl:4-7,14-18 and src/worker_bwt.erl:17, compiled into the kernels;
 * bwt_constants() lets the Erlang side assert they equal its own macros at load time. */
typedef struct {
    int32_t tolerance, min_len, max_range, max_len;       /* 20, 8, 45, 15 */
    int32_t match, mismatch, gap_penalty, gap_ext_penalty; /* 2, -1, -2, -1 */
    double unknown;                                        /* 1.6 */
    int32_t ref_extension_len;                             /* 200 */
} bwt_constants_t;
void bwt_constants(bwt_constants_t *out);

const char *bwt_last_error(void);

/* ------------------------------------------------------------------ execution options
 * Which of the library's equivalent kernels run is decided by this struct, never by the process environment (a library
 * loaded into a BEAM VM must not change behaviour with ambient variables).  Every field: -1 = the library's own choice;
 * results are bit-identical for every setting (tests/test_gpu_modes.py drives each one against the oracle).
 * bwt_options_init fills in -1 everywhere.  bwt_index_set_options: the defaults of every batch / pipe created on the index
 * afterwards and of the private batches of the host-buffer entry points (bwt_align_batch, ...); bwt_batch_set_options: one
 * batch.  Only bench.py, scripts/ and the tests translate environment variables into this struct. */
typedef struct {
    uint32_t struct_size;       /* sizeof(bwt_options): set by bwt_options_init */
    int32_t seed_fuse;          /* 0: the seed kernel does not cluster reads itself */
    int32_t seed_fast;          /* 0: general seed kernel only (no k_seed_fast) */
    int32_t seed_big;           /* 0 / 1: force the one- / two-register variant of k_seed_fast */
    int32_t search_ilp;         /* 2: two tasks per quad in k_search (table-less search) */
    int32_t sw_prefilter;       /* 0: every candidate goes to the DP kernels */
    int32_t sw_fuse_cand;       /* 0: separate k_candidates pass in front of the prefilter */
    int32_t sw_wave;            /* 0: no wave-per-alignment kernel for short DP queues */
    int32_t sw_band;            /* 0: no banded DP tiers (full rectangles only); 2: the 20- and 32-diagonal tiers only; else all three */
    int32_t sw_band_min;        /* DP queues up to this many candidates skip the banded tier (default 16384) */
    int32_t sw_lds;             /* 0: boundary rows in HBM slabs instead of LDS */
    int32_t sw_pk;              /* 0: no packed-i16 DP kernels; 2: without their wide pass (windows past 254 bases go to the generic kernel) */
    int32_t sw_trace_all;       /* 0: score-only DP pass first, traceback for what needs it */
    int32_t pf_blocks_per_cu, sw_blocks_per_cu, sw_waves_per_cu;   /* occupancy caps (experiments) */
    int32_t debug;              /* 1: diagnostic lines on stderr (hand-back reasons of k_seed_fast) */
    int32_t hits_narrow;        /* 1: the hit table is made in its NARROW form (bwt_batch_fetch_hits_narrow; a pipe's results then carry it) */
    int32_t sw_band12;          /* 0: no 12-diagonal tier in front of the banded DP tiers (square windows with <= 6 mismatches start at 20 diagonals) */
    int32_t up_zero_copy;       /* >= 1: a packed upload whose `codes` buffer is PINNED host memory (a pipe slot's is; the library asks
                                   hipPointerGetAttributes) is not copied: the unpack kernel reads it over the link itself, with 64
                                   workgroups (1) or with this many (> 1).  A pageable buffer is copied as always.  Off by default:
                                   DESIGN 8 has the measurement. */
    int32_t small_graph;        /* 0: bwt_small_align never replays its launches as a hipGraph (it does from the second batch of one
                                   shape in a row) */
    int32_t sw_cert;            /* 0: no mismatch-mask certificate (k_sw_cert) in front of the 12-diagonal tier: every candidate of its queue
                                   runs the banded DP; 2: only its 11-diagonal instance (square windows with up to 6 mismatches), not the
                                   19-diagonal one (7..10) */
    int32_t pf_kcap;            /* candidates of a read the fused prefilter pass tests itself before it lists the rest for the quad-per-candidate
                                   pass (default 1; 0: it lists them all) */
} bwt_options;
void bwt_options_init(bwt_options *o);

/* ------------------------------------------------------------------ index (a-11, worker init)
 * Replaces fm_index:get_index/2 (src/fm_index.erl:154-156) + the .ref load and 200-N halo of
 * src/worker_bwt.erl:48-55.  Parses the ETF .fm, keeps only L and SA (F and I are redundant,
 * SURVEY 8a-13; the identity I[r] = C0[L[r]] + occ(L[r],r) - 1 is verified while loading),
 * packs L into 64-byte rank blocks and uploads everything to `device`. */
typedef struct bwt_index bwt_index;
int bwt_index_open(const char *fm_path, const char *ref_path, int device, bwt_index **out);
/* Same, from arrays already in memory (monolithic indexes from bwt_build_index, tests).
 * L: n_rows bytes, row r (1-based) at L[r-1], '$' exactly once; SA: n_rows u32;
 * pointers = {Pc,Pg,Pt,Last}; ref: the chunk's .ref bytes (n_rows-1 of them). */
int bwt_index_from_arrays(const uint8_t *L, const uint32_t *SA, uint32_t n_rows,
                          const uint32_t pointers[4], int32_t shift,
                          const uint8_t *ref, uint64_t ref_len, int device, bwt_index **out);
void bwt_index_close(bwt_index *idx);
/* out = {Pc, Pg, Pt, Last, Shift}  (proplists pointers / shift of the Meta term) */
int bwt_index_meta(const bwt_index *idx, uint32_t out[5]);
/* bytes resident in HBM: {rank blocks, SA, ref1 + its bit-planes, suffix-interval table} */
int bwt_index_device_bytes(const bwt_index *idx, uint64_t out[4]);
/* Suffix-interval table: for every k-mer, k = 8..K, the (Sp, Ep, n) that the first k-1 steps of
 * bwa:find_seeds/5 (src/bwa.erl:31-45) reach -- computed once with the same LF step when the index is
 * opened, so that a task costs one table read instead of up to 14 dependent steps (K = 15 answers every
 * task; smaller K leaves 15-K steps).  Results are identical with or without it.
 * K: 0 = none (every task searched step by step), 8..15, <0 = default (by index size).
 * wide: 1 = the top level as 64-byte entries that also hold the first 14 SA values of the range, so that
 *   a task ending there needs no SA read and the seed kernel can cluster the read itself (64 * 4^K bytes:
 *   68.7 GB at K = 15); 2 = K = 15 only: also the LOWER levels where at least a quarter of the k-mers give a
 *   task that ends there a seed list (a read's short last sub-read ends below the top: 12 bases of a 100-bp
 *   read, 14 of a 150-bp one -- 1.1 GB for the 12-mers of a chromosome-sized index, 17 GB for the 14-mers of a
 *   genome-sized one); 0 = 8-byte entries throughout; <0 = default (2 when K = 15 and the top level leaves
 *   >= 32 GB of HBM free).
 * Not to be called while a batch on this index is running. */
int bwt_index_set_ktable(bwt_index *idx, int K, int wide);
/* current K (0 = none), + 16 when the top level is wide, + 1 << m for every lower level m (9..14) that is kept wide as well */
int bwt_index_ktable(const bwt_index *idx);
/* the same with the summary bitmap of the 8-byte levels chosen explicitly: summary 0 = none, 1 = where it pays (the
 * default), 2 = on every level (tests) */
int bwt_index_set_ktable_ex(bwt_index *idx, int K, int wide, int summary);
int bwt_index_set_options(bwt_index *idx, const bwt_options *opt);

/* ------------------------------------------------------------------ a-13 / 8f-1: index construction
 * Replaces fm_index:make_index/4 (src/fm_index.erl:127-141) = IUPAC map + random:uniform(4) for N
 * (AS183; rng_state in/out, NULL = the default process seed {3172,9814,20125}) + bwt:fm/1
 * (src/bwt.erl:84-120, suffix array by src/dsa.erl with '$' smallest) + fmi:get_index_pointers/1.
 * `ref` is the chunk as it goes into the .ref file (before substitution), `shift` the Meta value.
 * The suffix array is built on the GPU (prefix doubling + radix sort); the index stays in HBM. */
int bwt_build_index(const uint8_t *ref, uint64_t n, int32_t shift, uint32_t rng_state[3], int device,
                    bwt_index **out);
/* writes <fm_path> (ETF {Meta,FM}, byte-for-byte what term_to_binary/1 gives) and <ref_path> */
int bwt_index_save(const bwt_index *idx, const char *fm_path, const char *ref_path);

/* A compact native file next to the reference's format: the index exactly as it sits in HBM -- 64-byte rank
 * blocks, SA, the .ref bases -- behind a 128-byte header with the pointers, the shift and a checksum per section: 5.4 bytes
 * per row where the term format takes ~16 (a monolithic GRCh38 index would be a ~50 GB Erlang term).  The .fm / .ref pair
 * stays the interchange format with the reference; an index of any size can be kept and reopened this way.
 * bwt_native_check (host only, no GPU) verifies header and checksums and returns {Pc,Pg,Pt,Last,Shift}. */
int bwt_index_save_native(const bwt_index *idx, const char *path);
int bwt_index_open_native(const char *path, int device, bwt_index **out);
int bwt_native_check(const char *path, uint32_t out[5]);

/* copies the L column (n_rows bytes, '$' at its row) and the SA column back to the host */
int bwt_index_export(const bwt_index *idx, uint8_t *L, uint32_t *SA);

/* Host-only helpers on the on-disk format (no GPU needed):
 * bwt_fm_check parses a .fm exactly as bwt_index_open does (all identities verified) and returns
 * {Pc,Pg,Pt,Last,Shift}; bwt_fm_rewrite parses in_path and serialises it again the way
 * term_to_binary/1 does (src/fm_index.erl:140) -- byte-identical for files written by OTP 17. */
int bwt_fm_check(const char *fm_path, uint32_t out[5]);
int bwt_fm_rewrite(const char *in_path, const char *out_path);

/* ------------------------------------------------------------------ a-1..a-3
 * bwa:find_seeds/6 (src/bwa.erl:12-45, src/fmi.erl:25-39,89-100) on n sub-reads.
 * status[i]: 0 = seed list, 1 = no_seeds, 2 = too_many_seeds, -1 = the reference would crash
 * (last base not in ACGTN, src/bwa.erl:15-24) -- the one documented divergence: reported, not raised.
 * sp/ep/nmatch: final SA range (1-based rows, reference numbering) and matched length N.
 * seeds[seed_off[i] .. seed_off[i+1]) = SA[Ep]+N, SA[Ep-1]+N, ..., SA[Sp]+N  (the reference's list order). */
int bwt_find_seeds_batch(bwt_index *idx, const uint8_t *bases, const uint64_t *off, uint32_t n,
                         int32_t *status, uint32_t *sp, uint32_t *ep, uint32_t *nmatch,
                         uint64_t *seed_off /* n+1 */, int64_t *seeds, uint64_t seeds_cap,
                         uint64_t *seeds_needed);

/* ------------------------------------------------------------------ a-4..a-6
 * sga:sga/6 (src/sga.erl:17-76) on n reads -> CSR of {S,D} clusters in the reference's list
 * order (descending S).  read_status[i]: 0 ok, -1 reference would crash on this read. */
int bwt_sga_batch(bwt_index *idx, const uint8_t *bases, const uint64_t *off, uint32_t n,
                  uint64_t *clus_off /* n+1 */, int64_t *S, int32_t *D, uint64_t cap,
                  uint64_t *needed, int32_t *read_status);

/* ------------------------------------------------------------------ a-8..a-10
 * sw:sw/2 (src/sw.erl:47-202) on n (query, reference-window) pairs.
 * status[i]: 0 = {Score,Cigar}, 1 = no_match.  cigar text (no terminator) is
 * cigar[cig_off[i] .. cig_off[i+1]).  Needs no index. */
int bwt_sw_batch(int device, const uint8_t *q, const uint64_t *qoff, const uint8_t *r,
                 const uint64_t *roff, uint32_t n, int32_t *status, int32_t *score,
                 uint64_t *cig_off /* n+1 */, char *cigar, uint64_t cig_cap, uint64_t *cig_needed);

/* ------------------------------------------------------------------ a-7 (minus the final pick)
 * The `running` clause of worker_bwt:worker_loop/10 (src/worker_bwt.erl:73-106) for n reads:
 * sga -> window slice of Ref1 -> sw, returning for every read the `Cigars` list in the
 * reference's list order (ascending position); the final lists:sort/2 pick (:104) stays with
 * the caller (OTP decides ties).  read_status[i]:
 *    0  sga found clusters; hits[hit_off[i]..hit_off[i+1]) may still be empty (=> no_cigar)
 *    1  sga returned []: the reference emits nothing for this read (src/worker_bwt.erl:77)
 *   -1  reference would crash: bad last base;  -2  window outside Ref1 (src/worker_bwt.erl:89)
 * per hit: score, pos (= S - Ref_len + Shift), window offset into Ref1 and length (RefSeq),
 * cigar text at cigar[cig_off[h] .. cig_off[h+1]).  needed = {hits, cigar bytes}. */
int bwt_align_batch(bwt_index *idx, const uint8_t *bases, const uint64_t *off, uint32_t n,
                    int32_t *read_status, uint64_t *hit_off /* n+1 */,
                    int32_t *score, int64_t *pos, int64_t *win_off, int32_t *win_len,
                    uint64_t *cig_off /* hits+1 */, char *cigar,
                    uint64_t hit_cap, uint64_t cig_cap, uint64_t needed[2]);
/* Ref1 = 200 x 'N' ++ .ref ++ 200 x 'N' (src/worker_bwt.erl:54-55): copy [off, off+len) to out. */
int bwt_ref1_read(const bwt_index *idx, uint64_t off, uint64_t len, uint8_t *out);
/* the RefSeq of n hits in one go: windows [win_off[i], win_off[i]+win_len[i]) of Ref1, concatenated into out
 * (capacity out_cap bytes; out_off[n+1] receives the offsets; BWT_ENOSPC reports the total in *needed) */
int bwt_ref1_gather(const bwt_index *idx, const int64_t *win_off, const int32_t *win_len, uint64_t n,
                    uint8_t *out, uint64_t out_cap, uint64_t *out_off, uint64_t *needed);

/* ------------------------------------------------------------------ a-12: seeds.erl
 * seeds:generate_fs(Qsec, Len, Defects) (src/seeds.erl:11-85) writes the module `fs`: one matches/Len clause per way
 * of spending exactly Defects edits (mutation, insertion, deletion; edited positions are `_`) on the first Len+Defects
 * bases of Qsec, and fs:find_seeds(Ref_Seq), a greedy scan (hit at N -> record N, go on at N+Len; miss -> N+1) that
 * returns the 1-based hit positions newest first.
 *   bwt_seeds_generate_fs  the text of fs.erl, byte for byte what the generator's io:format calls print (host only);
 *   bwt_fs_find_seeds      what fs:find_seeds/1 of that module returns for `ref` -- on the GPU: the clause set is
 *                          evaluated as one small NFA per window start, the greedy scan replayed over the marked starts;
 *   bwt_fs_find_seeds_index the same over the bases of a resident index's chunk (the .ref bytes).
 * 2 <= Len <= 1024 (the generator itself crashes on 0 and 1), Defects <= 4, qlen >= Len+Defects (a shorter query
 * makes the generated file uncompilable), ref_len < 2^32-256.  BWT_ENOSPC reports the count in *needed. */
int bwt_seeds_generate_fs(const uint8_t *qsec, uint32_t qlen, uint32_t len, uint32_t defects, char *text,
                          uint64_t cap, uint64_t *needed);
int bwt_fs_find_seeds(int device, const uint8_t *qsec, uint32_t qlen, uint32_t len, uint32_t defects,
                      const uint8_t *ref, uint64_t ref_len, int64_t *pos, uint64_t cap, uint64_t *needed);
int bwt_fs_find_seeds_index(const bwt_index *idx, const uint8_t *qsec, uint32_t qlen, uint32_t len, uint32_t defects,
                            int64_t *pos, uint64_t cap, uint64_t *needed);

/* ------------------------------------------------------------------ 8f-2: FASTQ ingest
 * fastq:read_seqs/2 (src/fastq.erl:5-40) on a memory-mapped file, straight into the packed batch layout:
 * up to n records in the reference's list order (records are prepended => reverse file order);
 * Name = header line minus '@' and minus its last character, Seq = next line minus its last character.
 * *state: 0 = {ok,Acc}, 1 = eof (no record), 2 = {eof,Acc} (file exhausted before n records).
 * BWT_ENOSPC leaves the file position untouched and reports needed = {bases, name bytes}. */
typedef struct bwt_fastq bwt_fastq;
int bwt_fastq_open(const char *path, bwt_fastq **out);
void bwt_fastq_close(bwt_fastq *f);
/* Big batches are parsed by many threads (files whose records are four lines with an '@' header -- anything else falls back to
 * the sequential reader, which decides as fastq:read_seq/1 does).  threads: 0 = automatic, 1 = sequential only;
 * min_records: the batch size from which the threaded reader is used (default 65536). */
int bwt_fastq_set_threads(bwt_fastq *f, uint32_t threads, uint64_t min_records);
int bwt_fastq_read_seqs(bwt_fastq *f, uint32_t n, uint8_t *bases, uint64_t bases_cap, uint64_t *off /* n+1 */,
                        char *names, uint64_t names_cap, uint64_t *name_off /* n+1 */, uint32_t *n_out,
                        int *state, uint64_t needed[2]);

/* fastq:reverse_complement/1 (src/fastq.erl:43-50; exported by the reference, unused by its worker path) on n reads:
 * out receives the reverse complements at the same offsets (out may not alias bases).  status[i]: 0, or -1 where the
 * reference has no clause (a base outside ACGTN: function_clause) -- reported per read, that read's output is undefined. */
int bwt_reverse_complement_batch(int device, const uint8_t *bases, const uint64_t *off, uint32_t n, uint8_t *out,
                                 int32_t *status);

/* ------------------------------------------------------------------ resident-batch interface
 * The same pipeline split so that a caller can keep reads and hits in HBM (bench, multi-GPU
 * gather): upload once, run stages on the index's HIP stream, fetch what is wanted. */
typedef struct bwt_batch bwt_batch;
enum { BWT_STAGE_SEARCH = 1, BWT_STAGE_CLUSTER = 2, BWT_STAGE_EXTEND = 4 };
typedef struct {
    uint64_t n_reads, n_tasks;       /* sub-reads enumerated by skip_Ns */
    uint64_t lf_steps;               /* LF steps taken (two rank-block lookups each)        */
    uint64_t sa_reads;               /* SA entries gathered                                  */
    uint64_t n_clusters;             /* {S,D} candidates                                     */
    uint64_t sw_cells;               /* Lq x W summed over candidates                        */
    uint64_t n_hits, cigar_bytes;    /* alignments kept                                      */
    uint64_t window_bytes;           /* reference-window bytes read by the extend stage      */
    uint64_t n_float;                /* candidates that took the fp64 (N / 1.6) path         */
    uint64_t n_traced;               /* candidates that needed a real traceback (not "<Lq>M") */
    uint64_t n_dp;                   /* candidates the prefilter could not settle (ran the DP)  */
    uint64_t n_lookups;              /* tasks answered or started from the suffix-interval table */
    uint64_t n_small;                /* reads with 1..64 seeds left to k_cluster_small (the rest: none, fused, or wider kernels) */
    uint64_t n_handed_back;          /* reads k_seed_fast handed back to the general seed kernel (0 when it did not run) */
    uint64_t n_banded;               /* DP candidates settled by the banded kernel (exact by its certificate); n_dp counts them too */
    uint64_t n_certified;            /* candidates past the prefilter that k_sw_cert settled without a DP (square window, 3..6 mismatches on
                                        its diagonal, no other path can reach it: bwt_amd/csrc/sw_cert.h); n_dp does NOT count them */
} bwt_batch_counts;
int bwt_batch_create(bwt_index *idx, bwt_batch **out);
void bwt_batch_destroy(bwt_batch *b);
int bwt_batch_set_options(bwt_batch *b, const bwt_options *opt);
int bwt_batch_set_reads(bwt_batch *b, const uint8_t *bases, const uint64_t *off, uint32_t n);

/* ------------------------------------------------------------------ packed reads (the host boundary's wire format)
 * The reference hands a worker its reads as Erlang strings (src/master.erl:162-178 -> src/worker_bwt.erl:62-69); what has to
 * cross PCIe is two bits per base.  A packed batch is the concatenated base stream (read i = bases [off[i], off[i+1]), or
 * [i*read_len, (i+1)*read_len) when every read has the same length) as
 *   codes  two u32 per 32 bases: codes[2w] = bit i is the LOW bit of the 2-bit code (A C G T = 0 1 2 3) of base 32w+i,
 *          codes[2w+1] = the HIGH bits; a base outside ACGT has code 0.  bwt_packed_words(total) words, zero behind the
 *          last base.
 *   exc    one u64 per base outside ACGT, ascending by position: position | (uint64_t)byte << 56 (N, IUPAC codes,
 *          anything else -- the byte is kept because sw:sigma/2 compares read bytes, src/sw.erl:164-202).
 * 0.25 bytes per base up instead of 1 (+ 8 per read for offsets): 25 instead of 108 bytes for a 100 bp read.
 * bwt_pack_reads (host only, no GPU) packs an ASCII stream with `threads` host threads (0 = automatic); BWT_ENOSPC +
 * *n_exc = the count when exc_cap is too small (the codes are complete either way).
 * bwt_batch_set_reads_packed uploads a packed batch and expands it on the device into the bit-planes every kernel reads --
 * the same planes, bit for bit, that bwt_batch_set_reads derives from ASCII, so every result is identical.
 * read_len > 0: uniform batch (off may be NULL); read_len == 0: off[0..n] as in bwt_batch_set_reads (off[0] = 0). */
uint64_t bwt_packed_words(uint64_t total_bases);
int bwt_pack_reads(const uint8_t *bases, uint64_t total_bases, uint32_t *codes, uint64_t *exc, uint64_t exc_cap,
                   uint64_t *n_exc, uint32_t threads);
int bwt_batch_set_reads_packed(bwt_batch *b, const uint32_t *codes, uint64_t total_bases, uint32_t n, uint32_t read_len,
                               const uint64_t *off, const uint64_t *exc, uint64_t n_exc);
/* runs the requested stages (a later stage implies the earlier ones) and waits for them */
int bwt_batch_run(bwt_batch *b, int stages, bwt_batch_counts *counts);
/* per-kernel device time of the runs since the last reset, from hipEvents recorded on the
 * stream the kernels are launched on.  names: NUL-separated list; returns #kernels. */
int bwt_batch_profile(bwt_batch *b, int enable);
int bwt_batch_kernel_times(bwt_batch *b, char *names, uint32_t names_cap, double *ms, uint64_t *launches,
                           uint32_t cap);
/* device-to-device copy of the compact hit table {read:u32, score:i32, pos:i64} (16 B/hit)
 * into caller-owned device memory (e.g. a torch tensor used for the RCCL gather) */
int bwt_batch_hits_to_device(bwt_batch *b, void *dev_dst, uint64_t cap_hits);
/* the same with `read_base` added to every record's read index (slab-local -> global read ids for the gather) */
int bwt_batch_hits_to_device_global(bwt_batch *b, void *dev_dst, uint64_t cap_hits, uint32_t read_base);
int bwt_batch_fetch_tasks(bwt_batch *b, int32_t *status, uint32_t *sp, uint32_t *ep, uint32_t *nmatch,
                          uint32_t *task_read, uint32_t *task_end, uint64_t cap, uint64_t *needed);
int bwt_batch_fetch_clusters(bwt_batch *b, uint64_t *clus_off, int64_t *S, int32_t *D, uint64_t cap,
                             uint64_t *needed, int32_t *read_status);
int bwt_batch_fetch_hits(bwt_batch *b, int32_t *read_status, uint64_t *hit_off, int32_t *score,
                         int64_t *pos, int64_t *win_off, int32_t *win_len, uint64_t *cig_off,
                         char *cigar, uint64_t hit_cap, uint64_t cig_cap, uint64_t needed[2]);

/* ------------------------------------------------------------------ reverse strand and pairing
 * Functionality the reference does NOT have (its only trace is the unused fastq:reverse_complement/1, src/fastq.erl:43-50);
 * nothing changes unless these entry points are called.
 * bwt_batch_set_reads_both_strands: as bwt_batch_set_reads, and the batch then holds 2n reads -- read n + i is the reverse
 *   complement of read i (IUPAC codes complemented, made on the device).  Every stage and every fetch works on the 2n reads:
 *   a hit of read n + i is an alignment of read i's reverse strand (position and CIGAR are those of the reverse complement
 *   on the forward reference, as in SAM).
 * bwt_batch_pair_hits: after the extend stage of such a batch of 2 * n_pairs INTERLEAVED mates (reads 2p, 2p + 1): per pair
 *   the best proper combination -- one mate forward, the other reverse, the forward one first, fragment length (forward
 *   position to the end of the reverse hit on the reference) within min_insert..max_insert; highest score sum, the first in
 *   enumeration order (mate 1 forward first, then hits in table order) among equals.  hit1 / hit2 index the hit tables
 *   (0xffffffff: no proper combination); flags bit 0: mate 2 is the forward one. */
typedef struct { uint32_t hit1, hit2; int32_t insert, score; uint32_t n_proper, flags; } bwt_pair_rec;
int bwt_batch_set_reads_both_strands(bwt_batch *b, const uint8_t *bases, const uint64_t *off, uint32_t n);
int bwt_batch_pair_hits(bwt_batch *b, uint32_t n_pairs, int32_t min_insert, int32_t max_insert, bwt_pair_rec *pairs /* n_pairs */);

/* the hit tables as they sit in HBM, copied without any host-side reshaping: hits[h] = {read, score, pos} (the 16-byte
 * record that is also gathered over RCCL).  Any output pointer may be NULL (skipped); with pinned destination buffers
 * the copies run at full DMA speed.  BWT_ENOSPC + needed = {hits, cigar bytes} when a capacity is too small. */
typedef struct { uint32_t read; int32_t score; int64_t pos; } bwt_hit_rec;
int bwt_batch_fetch_hits_raw(bwt_batch *b, int32_t *read_status, uint64_t *hit_off, bwt_hit_rec *hits, int64_t *win_off,
                             int32_t *win_len, uint64_t *cig_off, char *cigar, uint64_t hit_cap, uint64_t cig_cap,
                             uint64_t needed[2]);

/* The compact hit table -- the device's own format, and what crosses PCIe and xGMI.  hits[] as above, ascending by read
 * (a read's hits adjacent, in the reference's list order); read_status one signed byte per read (values of bwt_align_batch).
 * Most hits are "plain": square window (RefSeq = Lq bases) and CIGAR "<Lq>M" -- for those nothing else is stored:
 *     win_len = Lq,  win_off = pos - Shift + 200 (src/worker_bwt.erl:86-94),  cigar = "<Lq>M".
 * Every other hit has a record in extra[] (ascending by hit): its window length and its CIGAR text
 * cigar[cig_off .. cig_off + cig_len).  needed = {hits, extras, cigar bytes}.  Any output pointer may be NULL. */
typedef struct { uint32_t hit, win_len, cig_off, cig_len; } bwt_hit_extra;
int bwt_batch_fetch_hits_compact(bwt_batch *b, int8_t *read_status, bwt_hit_rec *hits, bwt_hit_extra *extra, char *cigar,
                                 uint64_t hit_cap, uint64_t extra_cap, uint64_t cig_cap, uint64_t needed[3]);
/* The NARROW form of the compact table (bwt_options.hits_narrow = 1): most reads of a sequencing run have exactly one hit, and it is
 * plain -- for such a read nothing goes into hits[]: its read_status byte is 2 (instead of 0) and the k-th such read's hit is
 *     narrow_off[k]    window offset into Ref1 (u32: an index has fewer than 2^32 - 70000 rows), Pos = narrow_off - 200 + Shift
 *     narrow_score[k]  the score (i16: the form is only made for batches whose reads have at most 16383 bases; a batch of longer
 *                      reads simply has no narrow reads)
 * with win_len = Lq and cigar = "<Lq>M" as for every plain hit: 7 bytes per read down instead of 17.  hits[] / extra[] / cigar hold
 * everything else exactly as in the compact table (extra[].hit indexes hits[]).  needed = {narrow, hits, extras, cigar bytes}.
 * bwt_hits_expand_narrow is the decoder (host only); every entry point that hands out 16-byte records (bwt_batch_fetch_hits*,
 * bwt_batch_*_to_device, bwt_batch_pair_hits) first turns a narrow table back into that form on the device. */
int bwt_batch_fetch_hits_narrow(bwt_batch *b, int8_t *read_status, uint32_t *narrow_off, int16_t *narrow_score, bwt_hit_rec *hits,
                                bwt_hit_extra *extra, char *cigar, uint64_t narrow_cap, uint64_t hit_cap, uint64_t extra_cap, uint64_t cig_cap,
                                uint64_t needed[4]);
int bwt_hits_expand_narrow(uint32_t n, uint32_t read_len, const uint64_t *off, int32_t shift, const int8_t *read_status,
                           const uint32_t *narrow_off, const int16_t *narrow_score, uint64_t n_narrow, const bwt_hit_rec *hits, uint64_t n_hits,
                           const bwt_hit_extra *extra, uint64_t n_extra, const char *xcigar, uint64_t xcigar_len, uint64_t *hit_off,
                           int32_t *score, int64_t *pos, int64_t *win_off, int32_t *win_len, uint64_t *cig_off, char *cigar,
                           uint64_t cig_cap, uint64_t *cig_needed);
/* the same table device to device into caller-owned device memory (the send buffer of the RCCL hit gather): records with
 * `read_base` added to the read ids, extras and text as they are (hit indices and text offsets stay local to this batch: the
 * receiver adds the sender's bases).  Capacities in records / records / bytes. */
int bwt_batch_compact_to_device(bwt_batch *b, void *dst_hits, uint64_t cap_hits, void *dst_extra, uint64_t cap_extra,
                                void *dst_cigar, uint64_t cap_cigar, uint32_t read_base);
/* Host-side expansion of a compact table into the arrays of bwt_align_batch (no GPU): read_len > 0 for a uniform batch,
 * else off[0..n] gives the read lengths.  hit_off n+1, score/pos/win_off/win_len n_hits, cig_off n_hits+1; cigar
 * capacity cig_cap (BWT_ENOSPC + *cig_needed otherwise).  Every index of the table is checked (read ids < n and ascending,
 * extras ascending and naming hits, their text inside xcigar[0 .. xcigar_len)): BWT_EINVAL, never an access out of bounds. */
int bwt_hits_expand(uint32_t n, uint32_t read_len, const uint64_t *off, int32_t shift, const bwt_hit_rec *hits, uint64_t n_hits,
                    const bwt_hit_extra *extra, uint64_t n_extra, const char *xcigar, uint64_t xcigar_len, uint64_t *hit_off,
                    int32_t *score, int64_t *pos, int64_t *win_off, int32_t *win_len, uint64_t *cig_off, char *cigar,
                    uint64_t cig_cap, uint64_t *cig_needed);

/* ------------------------------------------------------------------ host-buffer pipeline
 * The worker's `running` clause as a pipeline over host buffers: `depth` slots, each with a resident batch (own HIP
 * stream), PINNED host buffers for the reads and the hit tables, and a host thread running
 *     H2D of the packed reads -> seed search + clustering + extension -> D2H of the compact hit table,
 * so that the upload of batch k+1, the kernels of batch k and the download of batch k-1 overlap (one upload, one batch's
 * kernels and one download at a time).  Per 100 bp read 25 bytes go up and ~17 come back.
 *   bwt_pipe_acquire_packed  a free slot's pinned input buffers for up to bases_cap bases / reads_cap reads / exc_cap
 *                     exceptions: the caller writes the packed batch (format above) straight into them -- or ASCII into
 *                     `ascii` (pageable staging of bases_cap bytes) and lets bwt_pipe_submit_ascii pack it; BWT_ENOSPC
 *                     when every slot is busy (take a result first).
 *   bwt_pipe_submit_packed   starts the slot (asynchronous); read_len / off as bwt_batch_set_reads_packed.
 *   bwt_pipe_submit_ascii    packs in->ascii[0 .. off[n]) with host threads into the slot's pinned buffers, then the same.
 *   bwt_pipe_acquire / bwt_pipe_submit   the ASCII form: pinned concatenated reads + n+1 offsets (off[0] = 0), uploaded
 *                     as they are (108 bytes per 100 bp read on the link).
 *   bwt_pipe_wait     blocks for the OLDEST submitted batch; the pointers in *res (pinned memory) stay valid until
 *                     bwt_pipe_release(slot), which makes the slot free again.  Until then bwt_pipe_batch(slot) is the
 *                     slot's resident batch, hit tables still in HBM (the RCCL gather copies them from there).
 * One caller thread per pipe (the handle is not re-entrant); several pipes may share an index. */
typedef struct bwt_pipe bwt_pipe;
typedef struct {
    uint32_t *codes;                 /* bwt_packed_words(bases_cap) u32 */
    uint64_t *exc;                   /* exc_cap */
    uint64_t *off;                   /* reads_cap + 1 (only read when read_len == 0) */
    uint8_t *ascii;                  /* bases_cap bytes of pageable staging for bwt_pipe_submit_ascii */
    uint64_t bases_cap, exc_cap;
    uint32_t reads_cap;
} bwt_pipe_input;
typedef struct {
    uint32_t slot, n_reads;
    int32_t status;                  /* 0 or the BWT_E* code of the batch */
    uint32_t read_len;               /* > 0: every read of the batch has this length */
    uint64_t n_hits, n_extra, cigar_bytes;
    const int8_t *read_status;       /* n_reads   (values as bwt_align_batch) */
    const bwt_hit_rec *hits;         /* n_hits, ascending by read */
    const bwt_hit_extra *extra;      /* n_extra, ascending by hit */
    const char *cigar;               /* cigar_bytes: the text of the extras */
    bwt_batch_counts counts;
    /* wall time of the slot's thread in its three stages (host packing excluded), and the time it waited for its turn */
    double ms_upload, ms_kernels, ms_download, ms_waiting;
    /* bwt_options.hits_narrow = 1 on the pipe: the reads with read_status 2 have their one plain hit here, not in hits[] (above) */
    uint64_t n_narrow;
    const uint32_t *narrow_off;      /* n_narrow */
    const int16_t *narrow_score;     /* n_narrow */
} bwt_pipe_result;
int bwt_pipe_create(bwt_index *idx, uint32_t depth /* 1..8 */, bwt_pipe **out);
void bwt_pipe_destroy(bwt_pipe *p);
int bwt_pipe_set_options(bwt_pipe *p, const bwt_options *opt);
/* per-kernel device time of the slots' batches (bwt_batch_profile / bwt_batch_kernel_times summed over the slots) */
int bwt_pipe_profile(bwt_pipe *p, int enable);
int bwt_pipe_kernel_times(bwt_pipe *p, char *names, uint32_t names_cap, double *ms, uint64_t *launches, uint32_t cap);
int bwt_pipe_acquire_packed(bwt_pipe *p, uint64_t bases_cap, uint32_t reads_cap, uint64_t exc_cap, bwt_pipe_input *in, uint32_t *slot);
int bwt_pipe_submit_packed(bwt_pipe *p, uint32_t slot, uint32_t n_reads, uint32_t read_len, uint64_t total_bases, uint64_t n_exc);
int bwt_pipe_submit_ascii(bwt_pipe *p, uint32_t slot, uint32_t n_reads, uint32_t read_len);
int bwt_pipe_acquire(bwt_pipe *p, uint64_t bases_cap, uint32_t reads_cap, uint8_t **bases, uint64_t **off, uint32_t *slot);
int bwt_pipe_submit(bwt_pipe *p, uint32_t slot, uint32_t n_reads);
int bwt_pipe_wait(bwt_pipe *p, bwt_pipe_result *res);
int bwt_pipe_release(bwt_pipe *p, uint32_t slot);
bwt_batch *bwt_pipe_batch(bwt_pipe *p, uint32_t slot);

/* ------------------------------------------------------------------ small batches: the reference's own message size
 * The master hands a worker 10 lists of 200 reads per message (src/master.erl:59) and worker_bwt runs its `running` clause once
 * per list (src/worker_bwt.erl:16,62-69,73-112).  A call of that size is all fixed cost on the general path (about forty launches,
 * six synchronisations, pageable copies, two-call sizing: 0.7-0.9 ms for 200-2000 reads).  bwt_small_align runs the SAME kernels
 * with what surrounds them removed: the reads go through pinned memory that the first kernel reads directly, no size comes back
 * to the host in mid-run (fixed capacities; queue lengths are read on the device), the compact hit table is written into
 * pinned memory by the last kernel, and one synchronisation ends the call.
 *   n <= BWT_SMALL_MAX_READS reads of up to ~200 bases, <= 1 M bases in all; a batch it does not take, or one that overflows a
 *   fixed capacity (more than 32768 candidates, 2^20 seeds left to the cluster kernels, a window wider than 2048 bases, a CIGAR
 *   longer than its slot), is run through the general path by the same call -- the results are identical either way and
 *   res->path says which ran: 1 = small path, 2 = small path overflowed and the general path redid it, 0 = general path.
 * The result is the compact hit table (format above; bwt_hits_expand gives the per-hit arrays); its pointers stay valid until the
 * next call on the handle.  One call at a time per handle; several handles may share an index. */
#define BWT_SMALL_MAX_READS 4096u
typedef struct bwt_small bwt_small;
typedef struct {
    uint32_t n_reads, path;
    uint64_t n_hits, n_extra, cigar_bytes;
    const int8_t *read_status;       /* n_reads   (values as bwt_align_batch) */
    const bwt_hit_rec *hits;         /* n_hits, ascending by read */
    const bwt_hit_extra *extra;      /* n_extra, ascending by hit */
    const char *cigar;               /* cigar_bytes: the text of the extras, each at its extra's cig_off / cig_len (NOT packed against
                                        each other here: a text takes the whole 16-byte pieces of its device slot) */
    uint64_t n_candidates, n_dp, n_float;   /* {S,D} candidates, those that ran a DP kernel, those that took the fp64 one */
    uint64_t overflow_flags;         /* path 2: why the small path gave up (2 a CIGAR longer than its slot, 16 a fixed capacity, ...) */
    double us_copy_in, us_enqueue, us_wait, us_total;   /* host clock since the call began: reads in pinned memory, kernels queued, results back, done */
    uint32_t graph;                  /* path 1: 0 the kernels were launched one by one, 1 they were replayed as one hipGraph (a batch of the
                                        same shape -- read count, lengths, options -- as the one before it), 2 the graph was captured in this call */
    uint32_t reserved_;
} bwt_small_result;
int bwt_small_create(bwt_index *idx, bwt_small **out);
void bwt_small_destroy(bwt_small *s);
int bwt_small_set_options(bwt_small *s, const bwt_options *opt);
int bwt_small_align(bwt_small *s, const uint8_t *bases, const uint64_t *off, uint32_t n, bwt_small_result *res);
/* per-kernel device time of the handle's runs (as bwt_batch_profile / bwt_batch_kernel_times) */
int bwt_small_profile(bwt_small *s, int enable);
int bwt_small_kernel_times(bwt_small *s, char *names, uint32_t names_cap, double *ms, uint64_t *launches, uint32_t cap);

/* ------------------------------------------------------------------ the whole job for one FASTQ file
 * What the reference's master and one worker do (src/master.erl:162-178 workload production, src/worker_bwt.erl:73-112 the
 * running clause incl. the final pick by lists:sort/2 with the comparator R1 > R2, src/master.erl:149-155 the result sink):
 * reads the FASTQ with the threaded reader straight into a bwt_pipe, picks per read, and writes to out_path one line
 *     "\n<Name>      <Chromosome>      <Pos>      <Cigar>      <Score>      <RefSeq>\n"
 * per cast whose score reaches `threshold` (master.erl:149 has 265), in the order the reference sends the casts.
 * out_path NULL: count only.  batch_reads: reads per GPU batch (0 = 2 M; rounded to whole workloads of 2000).
 * bwt_otp_pick: the index of the head of lists:sort(fun R1 > R2, L) for the scores of a read's candidates in list order
 * (OTP's merge sort decides among equal scores; restated, see bwt_amd/csrc/otp_sort.h). */
typedef struct {
    uint64_t reads, with_clusters, no_cigar, cigars, lines, crashes;
    double seconds;
} bwt_run_stats;
int bwt_run_fastq(bwt_index *idx, const char *fastq_path, const char *chromosome, int32_t threshold, uint32_t batch_reads,
                  const char *out_path, bwt_run_stats *stats);
int bwt_otp_pick(const int32_t *scores, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif

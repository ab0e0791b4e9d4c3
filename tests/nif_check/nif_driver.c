/* TEST INFRASTRUCTURE ONLY: drives c_src/bwt_nif.c through the stand-in NIF runtime (fake_enif.c) -- the first execution of the
 * shim's code.  What it does, and prints for tests/test_nif_exec.py to compare with libbwtgpu called directly:
 *     load/2, constants/0, index_open/3 on a missing file (an {error,_} term, no exception), index_open/3, index_meta/1,
 *     align_batch/2 with Seq as charlist / binary / nested iolist, on [] and on an improper list (badarg),
 *     align_workload/2 on the same reads in lists of 200 (must equal align_batch list by list),
 *     two threads calling align_batch on ONE Index term at once (the resource's mutex),
 *     then every environment is freed: the resource's destructor must have run exactly once and no term may be left.
 * usage: nif_driver <fm> <ref> <reads.tsv: name TAB seq per line> */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fake_enif.h"

static const ErlNifEntry *E;
static ERL_NIF_TERM call(ErlNifEnv *env, const char *name, int argc, const ERL_NIF_TERM *argv) {
    for (int i = 0; i < E->nfuncs; i++)
        if (!strcmp(E->funcs[i].name, name) && (int)E->funcs[i].arity == argc) return E->funcs[i].fptr(env, argc, argv);
    fprintf(stderr, "no such NIF: %s/%d\n", name, argc);
    exit(2);
}
typedef struct { char *name, *seq; } rd;
static rd *R;
static int NR;

static ERL_NIF_TERM seq_term(ErlNifEnv *env, const char *seq, int form) {
    const size_t n = strlen(seq);
    if (form == 1) return fake_make_binary(env, seq, n);
    if (form == 2 && n > 10)                                  /* an iolist: [<<first>>, "middle" | <<last>>] */
        return enif_make_list_cell(env, fake_make_binary(env, seq, 5),
                                   fake_make_improper(env, enif_make_string_len(env, seq + 5, n - 10, ERL_NIF_LATIN1), fake_make_binary(env, seq + n - 5, 5)));
    return enif_make_string_len(env, seq, n, ERL_NIF_LATIN1);        /* the reference's own form: a list of bytes */
}
static ERL_NIF_TERM qseq_list(ErlNifEnv *env, int lo, int hi) {
    ERL_NIF_TERM l = enif_make_list(env, 0);
    for (int i = hi; i-- > lo;)
        l = enif_make_list_cell(env, enif_make_tuple(env, 2, enif_make_string(env, R[i].name, ERL_NIF_LATIN1), seq_term(env, R[i].seq, i % 3)), l);
    return l;
}
static void print_results(const char *tag, ErlNifEnv *env, ERL_NIF_TERM res) {
    ERL_NIF_TERM h, t = res;
    if (fake_is_badarg(res)) { printf("%s BADARG\n", tag); return; }
    if (!enif_get_list_cell(env, t, &h, &t) && 0) return;
    t = res;
    unsigned n;
    if (!enif_get_list_length(env, res, &n)) { printf("%s ", tag); fake_print(stdout, res); printf("\n"); return; }
    while (enif_get_list_cell(env, t, &h, &t)) {
        const ERL_NIF_TERM *tp;
        int ar;
        enif_get_tuple(env, h, &ar, &tp);
        printf("%s ", tag); fake_print(stdout, tp[0]); printf(" "); fake_print(stdout, tp[1]); printf("\n");
    }
}
typedef struct { void *obj; int lo, hi; int id; char *out; size_t out_n; } targ;
static ErlNifResourceType *g_rt_probe;
static void *thread_main(void *a) {
    targ *ta = (targ *)a;
    ErlNifEnv *env = fake_env_new();
    ERL_NIF_TERM argv[2] = {enif_make_resource(env, ta->obj), qseq_list(env, ta->lo, ta->hi)};     /* (a copy of the Index term in this process) */
    ERL_NIF_TERM res = call(env, "align_batch", 2, argv);
    FILE *f = open_memstream(&ta->out, &ta->out_n);
    ERL_NIF_TERM h, t = res;
    while (enif_get_list_cell(env, t, &h, &t)) {
        const ERL_NIF_TERM *tp; int ar;
        enif_get_tuple(env, h, &ar, &tp);
        fprintf(f, "T "); fake_print(f, tp[0]); fprintf(f, " "); fake_print(f, tp[1]); fprintf(f, "\n");
    }
    fclose(f);
    fake_env_free(env);
    return NULL;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: nif_driver fm ref reads.tsv\n"); return 2; }
    FILE *f = fopen(argv[3], "r");
    if (!f) { perror(argv[3]); return 2; }
    char *line = NULL; size_t cap = 0; ssize_t len;
    int capr = 0;
    while ((len = getline(&line, &cap, f)) > 0) {
        if (line[len - 1] == '\n') line[--len] = 0;
        char *tab = strchr(line, '\t');
        if (!tab) continue;
        *tab = 0;
        if (NR == capr) { capr = capr ? 2 * capr : 1024; R = realloc(R, sizeof(rd) * capr); }
        R[NR].name = strdup(line); R[NR].seq = strdup(tab + 1); NR++;
    }
    free(line); fclose(f);
    E = nif_init();
    ErlNifEnv *env = fake_env_new();
    void *priv = NULL;
    if (E->load(env, &priv, enif_make_int(env, 0)) != 0) { printf("LOAD failed\n"); return 1; }
    printf("MODULE %s %d\n", E->name, E->nfuncs);
    printf("CONSTANTS "); fake_print(stdout, call(env, "constants", 0, NULL)); printf("\n");
    {
        ERL_NIF_TERM a[3] = {enif_make_string(env, "/nonexistent/x.fm", ERL_NIF_LATIN1), enif_make_string(env, "/nonexistent/x.ref", ERL_NIF_LATIN1), enif_make_int(env, 0)};
        ERL_NIF_TERM r = call(env, "index_open", 3, a);
        const ERL_NIF_TERM *tp; int ar;
        printf("OPEN_MISSING %s\n", enif_get_tuple(env, r, &ar, &tp) && ar == 2 ? "error_tuple" : "UNEXPECTED");
        ERL_NIF_TERM b[3] = {enif_make_atom(env, "not_a_string"), a[1], a[2]};
        printf("OPEN_BADARG %s\n", fake_is_badarg(call(env, "index_open", 3, b)) ? "badarg" : "UNEXPECTED");
    }
    ERL_NIF_TERM oa[3] = {enif_make_string(env, argv[1], ERL_NIF_LATIN1), enif_make_string(env, argv[2], ERL_NIF_LATIN1), enif_make_int(env, 0)};
    ERL_NIF_TERM opened = call(env, "index_open", 3, oa);
    const ERL_NIF_TERM *otp; int oar;
    if (!enif_get_tuple(env, opened, &oar, &otp) || oar != 2) { printf("OPEN failed\n"); return 1; }
    printf("OPEN "); fake_print(stdout, otp[0]); printf("\n");
    ERL_NIF_TERM Index = otp[1];
    printf("META "); fake_print(stdout, call(env, "index_meta", 1, &Index)); printf("\n");
    printf("META_BADARG %s\n", fake_is_badarg(call(env, "index_meta", 1, &oa[2])) ? "badarg" : "UNEXPECTED");
    {   /* one call with every read */
        ERL_NIF_TERM a[2] = {Index, qseq_list(env, 0, NR)};
        print_results("R", env, call(env, "align_batch", 2, a));
    }
    {   /* [] and things that are not QseqLists */
        ERL_NIF_TERM a[2] = {Index, enif_make_list(env, 0)};
        printf("EMPTY "); fake_print(stdout, call(env, "align_batch", 2, a)); printf("\n");
        ERL_NIF_TERM imp[2] = {Index, fake_make_improper(env, enif_make_tuple(env, 2, enif_make_string(env, "x", ERL_NIF_LATIN1), enif_make_string(env, "ACGT", ERL_NIF_LATIN1)), enif_make_atom(env, "tail"))};
        printf("IMPROPER %s\n", fake_is_badarg(call(env, "align_batch", 2, imp)) ? "badarg" : "UNEXPECTED");
        ERL_NIF_TERM bad[2] = {Index, enif_make_list(env, 1, enif_make_atom(env, "not_a_pair"))};
        printf("NOT_A_PAIR %s\n", fake_is_badarg(call(env, "align_batch", 2, bad)) ? "badarg" : "UNEXPECTED");
        ERL_NIF_TERM bad2[2] = {Index, enif_make_list(env, 1, enif_make_tuple(env, 2, enif_make_string(env, "x", ERL_NIF_LATIN1), enif_make_atom(env, "no_seq")))};
        printf("NO_SEQ %s\n", fake_is_badarg(call(env, "align_batch", 2, bad2)) ? "badarg" : "UNEXPECTED");
        ERL_NIF_TERM bad3[2] = {enif_make_int(env, 5), enif_make_list(env, 0)};
        printf("NO_INDEX %s\n", fake_is_badarg(call(env, "align_batch", 2, bad3)) ? "badarg" : "UNEXPECTED");
    }
    {   /* a whole workload message: lists of 200 (+ an empty one) */
        ERL_NIF_TERM wl = enif_make_list(env, 1, enif_make_list(env, 0));
        int nl = (NR + 199) / 200;
        for (int k = nl; k-- > 0;) wl = enif_make_list_cell(env, qseq_list(env, 200 * k, 200 * (k + 1) < NR ? 200 * (k + 1) : NR), wl);
        ERL_NIF_TERM a[2] = {Index, wl};
        ERL_NIF_TERM res = call(env, "align_workload", 2, a), h, t = res;
        int k = 0;
        while (enif_get_list_cell(env, t, &h, &t)) { char tag[32]; snprintf(tag, sizeof tag, "W%d", k++); print_results(tag, env, h); if (!fake_is_badarg(h)) { unsigned n; enif_get_list_length(env, h, &n); printf("WLEN %d %u\n", k - 1, n); } }
        ERL_NIF_TERM bw[2] = {Index, enif_make_list(env, 1, enif_make_atom(env, "x"))};
        printf("WORKLOAD_BADARG %s\n", fake_is_badarg(call(env, "align_workload", 2, bw)) ? "badarg" : "UNEXPECTED");
    }
    {   /* two processes, one Index term */
        void *obj = NULL;
        (void)g_rt_probe;
        extern int fake_resource_object(ERL_NIF_TERM t, void **obj);
        fake_resource_object(Index, &obj);
        pthread_t th[2];
        targ ta[2] = {{obj, 0, NR / 2, 0, NULL, 0}, {obj, NR / 2, NR, 1, NULL, 0}};
        for (int i = 0; i < 2; i++) pthread_create(&th[i], NULL, thread_main, &ta[i]);
        for (int i = 0; i < 2; i++) pthread_join(th[i], NULL);
        for (int i = 0; i < 2; i++) { fwrite(ta[i].out, 1, ta[i].out_n, stdout); free(ta[i].out); }
    }
    printf("BEFORE_FREE resources %ld dtors %ld\n", fake_live_resources(), fake_dtor_calls());
    fake_env_free(env);                                        /* the last reference to the Index goes: destructor -> bwt_index_close */
    printf("AFTER_FREE resources %ld dtors %ld terms %ld\n", fake_live_resources(), fake_dtor_calls(), fake_live_terms());
    for (int i = 0; i < NR; i++) { free(R[i].name); free(R[i].seq); }
    free(R);
    printf("DONE\n");
    return 0;
}

/* TEST INFRASTRUCTURE ONLY: see fake_enif.h. */
#include "fake_enif.h"
#include <pthread.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

typedef enum { T_INT, T_DOUBLE, T_ATOM, T_TUPLE, T_NIL, T_CONS, T_BINARY, T_RESOURCE, T_BADARG } tag_t;
struct res_hdr { ErlNifResourceType *type; int refc; pthread_mutex_t mu; };
struct term {
    tag_t tag;
    union {
        long long i; double d; char *atom;
        struct { int n; ERL_NIF_TERM *e; } tup;
        struct { ERL_NIF_TERM head, tail; } cons;
        struct { size_t n; unsigned char *p; } bin;
        struct res_hdr *res;
    } u;
    struct term *next;
};
struct enif_environment_t { struct term *all; void **scratch; size_t n_scratch; };
struct enif_resource_type_t { char name[64]; ErlNifResourceDtor *dtor; };
struct ErlNifMutex_ { pthread_mutex_t m; };

static long g_terms, g_res, g_dtors;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static void count(long *c, long d) { pthread_mutex_lock(&g_mu); *c += d; pthread_mutex_unlock(&g_mu); }
long fake_live_terms(void) { return g_terms; }
long fake_live_resources(void) { return g_res; }
long fake_dtor_calls(void) { return g_dtors; }

static struct term *T(ERL_NIF_TERM t) { return (struct term *)t; }
static ERL_NIF_TERM mk(ErlNifEnv *env, tag_t tag) {
    struct term *t = calloc(1, sizeof *t);
    if (!t) abort();
    t->tag = tag; t->next = env->all; env->all = t;
    count(&g_terms, 1);
    return (ERL_NIF_TERM)t;
}
ErlNifEnv *fake_env_new(void) { return calloc(1, sizeof(ErlNifEnv)); }
static void res_release(struct res_hdr *h) {
    pthread_mutex_lock(&h->mu);
    const int left = --h->refc;
    pthread_mutex_unlock(&h->mu);
    if (left == 0) {
        if (h->type->dtor) { h->type->dtor(NULL, (void *)(h + 1)); count(&g_dtors, 1); }
        pthread_mutex_destroy(&h->mu);
        free(h);
        count(&g_res, -1);
    }
}
void fake_env_free(ErlNifEnv *env) {
    struct term *t = env->all;
    while (t) {
        struct term *nx = t->next;
        if (t->tag == T_ATOM) free(t->u.atom);
        else if (t->tag == T_TUPLE) free(t->u.tup.e);
        else if (t->tag == T_BINARY) free(t->u.bin.p);
        else if (t->tag == T_RESOURCE) res_release(t->u.res);
        free(t);
        count(&g_terms, -1);
        t = nx;
    }
    for (size_t i = 0; i < env->n_scratch; i++) free(env->scratch[i]);
    free(env->scratch);
    free(env);
}
static void *scratch(ErlNifEnv *env, size_t n) {
    void *p = malloc(n ? n : 1);
    env->scratch = realloc(env->scratch, sizeof(void *) * (env->n_scratch + 1));
    if (!p || !env->scratch) abort();
    env->scratch[env->n_scratch++] = p;
    return p;
}

/* ---- resources */
ErlNifResourceType *enif_open_resource_type(ErlNifEnv *env, const char *module_str, const char *name, ErlNifResourceDtor *dtor,
                                            ErlNifResourceFlags flags, ErlNifResourceFlags *tried) {
    (void)env; (void)module_str; (void)flags;
    ErlNifResourceType *rt = calloc(1, sizeof *rt);           /* (lives as long as the process: a loaded module's type) */
    if (!rt) return NULL;
    strncpy(rt->name, name, sizeof rt->name - 1);
    rt->dtor = dtor;
    if (tried) *tried = ERL_NIF_RT_CREATE;
    return rt;
}
void *enif_alloc_resource(ErlNifResourceType *type, size_t size) {
    struct res_hdr *h = calloc(1, sizeof *h + size);
    if (!h) return NULL;
    h->type = type; h->refc = 1;
    pthread_mutex_init(&h->mu, NULL);
    count(&g_res, 1);
    return (void *)(h + 1);
}
void enif_release_resource(void *obj) { res_release((struct res_hdr *)obj - 1); }
ERL_NIF_TERM enif_make_resource(ErlNifEnv *env, void *obj) {
    struct res_hdr *h = (struct res_hdr *)obj - 1;
    pthread_mutex_lock(&h->mu); h->refc++; pthread_mutex_unlock(&h->mu);
    ERL_NIF_TERM t = mk(env, T_RESOURCE);
    T(t)->u.res = h;
    return t;
}
int enif_get_resource(ErlNifEnv *env, ERL_NIF_TERM term, ErlNifResourceType *type, void **objp) {
    (void)env;
    if (T(term)->tag != T_RESOURCE || T(term)->u.res->type != type) return 0;
    *objp = (void *)(T(term)->u.res + 1);
    return 1;
}

/* ---- reading terms */
int enif_get_int(ErlNifEnv *env, ERL_NIF_TERM term, int *ip) {
    (void)env;
    if (T(term)->tag != T_INT || T(term)->u.i < -2147483648LL || T(term)->u.i > 2147483647LL) return 0;
    *ip = (int)T(term)->u.i;
    return 1;
}
int enif_get_list_length(ErlNifEnv *env, ERL_NIF_TERM term, unsigned *len) {
    (void)env;
    unsigned n = 0;
    while (T(term)->tag == T_CONS) { n++; term = T(term)->u.cons.tail; }
    if (T(term)->tag != T_NIL) return 0;                      /* improper list */
    *len = n;
    return 1;
}
int enif_get_list_cell(ErlNifEnv *env, ERL_NIF_TERM term, ERL_NIF_TERM *head, ERL_NIF_TERM *tail) {
    (void)env;
    if (T(term)->tag != T_CONS) return 0;
    *head = T(term)->u.cons.head; *tail = T(term)->u.cons.tail;
    return 1;
}
int enif_get_tuple(ErlNifEnv *env, ERL_NIF_TERM tpl, int *arity, const ERL_NIF_TERM **array) {
    (void)env;
    if (T(tpl)->tag != T_TUPLE) return 0;
    *arity = T(tpl)->u.tup.n; *array = T(tpl)->u.tup.e;
    return 1;
}
int enif_get_string(ErlNifEnv *env, ERL_NIF_TERM list, char *buf, unsigned len, ErlNifCharEncoding enc) {
    (void)env; (void)enc;
    unsigned n = 0;
    while (T(list)->tag == T_CONS) {
        struct term *h = T(T(list)->u.cons.head);
        if (h->tag != T_INT || h->u.i < 0 || h->u.i > 255) return 0;
        if (n + 1 >= len) { if (len) buf[len - 1] = 0; return -(int)len; }          /* truncated */
        buf[n++] = (char)h->u.i;
        list = T(list)->u.cons.tail;
    }
    if (T(list)->tag != T_NIL || !len) return 0;
    buf[n] = 0;
    return (int)n + 1;
}
static int iolist_size(ERL_NIF_TERM t, size_t *n) {
    struct term *x = T(t);
    if (x->tag == T_BINARY) { *n += x->u.bin.n; return 1; }
    if (x->tag == T_NIL) return 1;
    if (x->tag != T_CONS) return 0;
    while (x->tag == T_CONS) {
        struct term *h = T(x->u.cons.head);
        if (h->tag == T_INT) { if (h->u.i < 0 || h->u.i > 255) return 0; *n += 1; }
        else if (!iolist_size(x->u.cons.head, n)) return 0;
        x = T(x->u.cons.tail);
    }
    if (x->tag == T_BINARY) { *n += x->u.bin.n; return 1; }    /* a binary tail is allowed in an iolist */
    return x->tag == T_NIL;
}
static void iolist_copy(ERL_NIF_TERM t, unsigned char **p) {
    struct term *x = T(t);
    if (x->tag == T_BINARY) { memcpy(*p, x->u.bin.p, x->u.bin.n); *p += x->u.bin.n; return; }
    while (x->tag == T_CONS) {
        struct term *h = T(x->u.cons.head);
        if (h->tag == T_INT) *(*p)++ = (unsigned char)h->u.i; else iolist_copy(x->u.cons.head, p);
        x = T(x->u.cons.tail);
    }
    if (x->tag == T_BINARY) { memcpy(*p, x->u.bin.p, x->u.bin.n); *p += x->u.bin.n; }
}
int enif_inspect_iolist_as_binary(ErlNifEnv *env, ERL_NIF_TERM term, ErlNifBinary *bin) {
    size_t n = 0;
    if (!iolist_size(term, &n)) return 0;
    if (T(term)->tag == T_BINARY) { bin->size = T(term)->u.bin.n; bin->data = T(term)->u.bin.p; return 1; }
    unsigned char *buf = scratch(env, n), *p = buf;           /* (ERTS: a temporary binary that lives until the NIF returns) */
    iolist_copy(term, &p);
    bin->size = n; bin->data = buf;
    return 1;
}

/* ---- making terms */
ERL_NIF_TERM enif_make_atom(ErlNifEnv *env, const char *name) {
    ERL_NIF_TERM t = mk(env, T_ATOM);
    T(t)->u.atom = strdup(name);
    return t;
}
ERL_NIF_TERM enif_make_int(ErlNifEnv *env, int i) { ERL_NIF_TERM t = mk(env, T_INT); T(t)->u.i = i; return t; }
ERL_NIF_TERM enif_make_uint(ErlNifEnv *env, unsigned i) { ERL_NIF_TERM t = mk(env, T_INT); T(t)->u.i = i; return t; }
ERL_NIF_TERM enif_make_int64(ErlNifEnv *env, ErlNifSInt64 i) { ERL_NIF_TERM t = mk(env, T_INT); T(t)->u.i = i; return t; }
ERL_NIF_TERM enif_make_double(ErlNifEnv *env, double d) { ERL_NIF_TERM t = mk(env, T_DOUBLE); T(t)->u.d = d; return t; }
ERL_NIF_TERM enif_make_list_cell(ErlNifEnv *env, ERL_NIF_TERM car, ERL_NIF_TERM cdr) {
    ERL_NIF_TERM t = mk(env, T_CONS);
    T(t)->u.cons.head = car; T(t)->u.cons.tail = cdr;
    return t;
}
ERL_NIF_TERM fake_make_improper(ErlNifEnv *env, ERL_NIF_TERM head, ERL_NIF_TERM tail) { return enif_make_list_cell(env, head, tail); }
ERL_NIF_TERM enif_make_string_len(ErlNifEnv *env, const char *string, size_t len, ErlNifCharEncoding enc) {
    (void)enc;
    ERL_NIF_TERM l = mk(env, T_NIL);
    for (size_t i = len; i-- > 0;) l = enif_make_list_cell(env, enif_make_int(env, (unsigned char)string[i]), l);
    return l;
}
ERL_NIF_TERM enif_make_string(ErlNifEnv *env, const char *string, ErlNifCharEncoding enc) { return enif_make_string_len(env, string, strlen(string), enc); }
ERL_NIF_TERM enif_make_tuple(ErlNifEnv *env, unsigned cnt, ...) {
    ERL_NIF_TERM t = mk(env, T_TUPLE);
    T(t)->u.tup.n = (int)cnt;
    T(t)->u.tup.e = malloc(sizeof(ERL_NIF_TERM) * (cnt ? cnt : 1));
    va_list ap;
    va_start(ap, cnt);
    for (unsigned i = 0; i < cnt; i++) T(t)->u.tup.e[i] = va_arg(ap, ERL_NIF_TERM);
    va_end(ap);
    return t;
}
ERL_NIF_TERM enif_make_list(ErlNifEnv *env, unsigned cnt, ...) {
    ERL_NIF_TERM *e = malloc(sizeof(ERL_NIF_TERM) * (cnt ? cnt : 1));
    va_list ap;
    va_start(ap, cnt);
    for (unsigned i = 0; i < cnt; i++) e[i] = va_arg(ap, ERL_NIF_TERM);
    va_end(ap);
    ERL_NIF_TERM l = mk(env, T_NIL);
    for (unsigned i = cnt; i-- > 0;) l = enif_make_list_cell(env, e[i], l);
    free(e);
    return l;
}
ERL_NIF_TERM enif_make_badarg(ErlNifEnv *env) { return mk(env, T_BADARG); }
ERL_NIF_TERM fake_make_binary(ErlNifEnv *env, const void *data, size_t n) {
    ERL_NIF_TERM t = mk(env, T_BINARY);
    T(t)->u.bin.n = n; T(t)->u.bin.p = malloc(n ? n : 1);
    if (n) memcpy(T(t)->u.bin.p, data, n);
    return t;
}
int fake_is_badarg(ERL_NIF_TERM t) { return T(t)->tag == T_BADARG; }

/* ---- mutex */
ErlNifMutex *enif_mutex_create(char *name) {
    (void)name;
    ErlNifMutex *m = malloc(sizeof *m);
    if (m) pthread_mutex_init(&m->m, NULL);
    return m;
}
void enif_mutex_destroy(ErlNifMutex *m) { pthread_mutex_destroy(&m->m); free(m); }
void enif_mutex_lock(ErlNifMutex *m) { pthread_mutex_lock(&m->m); }
void enif_mutex_unlock(ErlNifMutex *m) { pthread_mutex_unlock(&m->m); }

/* ---- printing */
static int printable_string(ERL_NIF_TERM t) {
    int n = 0;
    while (T(t)->tag == T_CONS) {
        struct term *h = T(T(t)->u.cons.head);
        if (h->tag != T_INT || h->u.i < 32 || h->u.i > 126 || h->u.i == '"') return 0;
        n++; t = T(t)->u.cons.tail;
    }
    return T(t)->tag == T_NIL && n > 0;
}
void fake_print(FILE *f, ERL_NIF_TERM t) {
    struct term *x = T(t);
    switch (x->tag) {
    case T_INT: fprintf(f, "%lld", x->u.i); break;
    case T_DOUBLE: fprintf(f, "%g", x->u.d); break;
    case T_ATOM: fprintf(f, "%s", x->u.atom); break;
    case T_NIL: fprintf(f, "[]"); break;
    case T_BADARG: fprintf(f, "BADARG"); break;
    case T_RESOURCE: fprintf(f, "#Ref<%s>", x->u.res->type->name); break;
    case T_BINARY: fprintf(f, "<<\"%.*s\">>", (int)x->u.bin.n, (const char *)x->u.bin.p); break;
    case T_TUPLE:
        fputc('{', f);
        for (int i = 0; i < x->u.tup.n; i++) { if (i) fputc(',', f); fake_print(f, x->u.tup.e[i]); }
        fputc('}', f);
        break;
    case T_CONS:
        if (printable_string(t)) {
            fputc('"', f);
            for (; T(t)->tag == T_CONS; t = T(t)->u.cons.tail) fputc((int)T(T(t)->u.cons.head)->u.i, f);
            fputc('"', f);
        } else {
            fputc('[', f);
            int first = 1;
            for (; T(t)->tag == T_CONS; t = T(t)->u.cons.tail) { if (!first) fputc(',', f); first = 0; fake_print(f, T(t)->u.cons.head); }
            if (T(t)->tag != T_NIL) { fputc('|', f); fake_print(f, t); }
            fputc(']', f);
        }
        break;
    }
}
/* the driver's way to hand one resource object to a second "process" (what enif_make_copy of the term does in ERTS) */
int fake_resource_object(ERL_NIF_TERM t, void **obj) {
    if (T(t)->tag != T_RESOURCE) return 0;
    *obj = (void *)(T(t)->u.res + 1);
    return 1;
}

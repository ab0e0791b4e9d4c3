/* TEST INFRASTRUCTURE ONLY -- NOT OTP's erl_nif.h.
 * Declarations (no definitions) of the subset of the Erlang NIF API that c_src/bwt_nif.c uses, written from the
 * published erl_nif documentation, so that tests/test_nif_shim.py can type-check the shim with `gcc -fsyntax-only`
 * in a container that has no Erlang/OTP.  Nothing links against this; a real build uses $ERL_ROOT/usr/include. */
#ifndef BWT_TEST_ERL_NIF_DECLS_H
#define BWT_TEST_ERL_NIF_DECLS_H
#include <stddef.h>
typedef unsigned long ERL_NIF_TERM;
typedef long ErlNifSInt64;
typedef struct enif_environment_t ErlNifEnv;
typedef struct enif_resource_type_t ErlNifResourceType;
typedef struct ErlNifMutex_ ErlNifMutex;
typedef struct { size_t size; unsigned char *data; void *ref_bin; void *spare[2]; } ErlNifBinary;
typedef void ErlNifResourceDtor(ErlNifEnv *, void *);
typedef enum { ERL_NIF_RT_CREATE = 1, ERL_NIF_RT_TAKEOVER = 2 } ErlNifResourceFlags;
typedef enum { ERL_NIF_LATIN1 = 1 } ErlNifCharEncoding;
typedef struct {
    const char *name; unsigned arity;
    ERL_NIF_TERM (*fptr)(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]);
    unsigned flags;
} ErlNifFunc;
#define ERL_NIF_DIRTY_JOB_CPU_BOUND 1
#define ERL_NIF_DIRTY_JOB_IO_BOUND 2
ErlNifResourceType *enif_open_resource_type(ErlNifEnv *, const char *module_str, const char *name, ErlNifResourceDtor *dtor,
                                            ErlNifResourceFlags flags, ErlNifResourceFlags *tried);
void *enif_alloc_resource(ErlNifResourceType *type, size_t size);
void enif_release_resource(void *obj);
ERL_NIF_TERM enif_make_resource(ErlNifEnv *, void *obj);
int enif_get_resource(ErlNifEnv *, ERL_NIF_TERM term, ErlNifResourceType *type, void **objp);
int enif_get_string(ErlNifEnv *, ERL_NIF_TERM list, char *buf, unsigned len, ErlNifCharEncoding);
int enif_get_int(ErlNifEnv *, ERL_NIF_TERM term, int *ip);
int enif_get_list_length(ErlNifEnv *, ERL_NIF_TERM term, unsigned *len);
int enif_get_list_cell(ErlNifEnv *, ERL_NIF_TERM term, ERL_NIF_TERM *head, ERL_NIF_TERM *tail);
int enif_get_tuple(ErlNifEnv *, ERL_NIF_TERM tpl, int *arity, const ERL_NIF_TERM **array);
int enif_inspect_iolist_as_binary(ErlNifEnv *, ERL_NIF_TERM term, ErlNifBinary *bin);
ERL_NIF_TERM enif_make_atom(ErlNifEnv *, const char *name);
ERL_NIF_TERM enif_make_int(ErlNifEnv *, int i);
ERL_NIF_TERM enif_make_uint(ErlNifEnv *, unsigned i);
ERL_NIF_TERM enif_make_int64(ErlNifEnv *, ErlNifSInt64);
ERL_NIF_TERM enif_make_double(ErlNifEnv *, double d);
ERL_NIF_TERM enif_make_string(ErlNifEnv *, const char *string, ErlNifCharEncoding);
ERL_NIF_TERM enif_make_string_len(ErlNifEnv *, const char *string, size_t len, ErlNifCharEncoding);
ERL_NIF_TERM enif_make_tuple(ErlNifEnv *, unsigned cnt, ...);
ERL_NIF_TERM enif_make_list(ErlNifEnv *, unsigned cnt, ...);
ERL_NIF_TERM enif_make_list_cell(ErlNifEnv *, ERL_NIF_TERM car, ERL_NIF_TERM cdr);
ERL_NIF_TERM enif_make_badarg(ErlNifEnv *);
#define enif_make_tuple2(env, e1, e2) enif_make_tuple(env, 2, e1, e2)
#define enif_make_tuple3(env, e1, e2, e3) enif_make_tuple(env, 3, e1, e2, e3)
#define enif_make_tuple5(env, e1, e2, e3, e4, e5) enif_make_tuple(env, 5, e1, e2, e3, e4, e5)
ErlNifMutex *enif_mutex_create(char *name);
void enif_mutex_destroy(ErlNifMutex *);
void enif_mutex_lock(ErlNifMutex *);
void enif_mutex_unlock(ErlNifMutex *);
typedef struct { const char *name; int nfuncs; ErlNifFunc *funcs; int (*load)(ErlNifEnv *, void **, ERL_NIF_TERM); } ErlNifEntry;
const ErlNifEntry *nif_init(void);
#define ERL_NIF_INIT(NAME, FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD)                                                   \
    const ErlNifEntry *nif_init(void) {                                                                            \
        static const ErlNifEntry e = {#NAME, sizeof(FUNCS) / sizeof(FUNCS[0]), FUNCS, LOAD};                       \
        (void)(RELOAD); (void)(UPGRADE); (void)(UNLOAD);                                                           \
        return &e;                                                                                                 \
    }
#endif

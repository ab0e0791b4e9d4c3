/* TEST INFRASTRUCTURE ONLY -- a small stand-in for the part of the Erlang NIF runtime that c_src/bwt_nif.c uses, so that the
 * shim's code (term marshalling, resource lifetime, mutex, error paths) can be EXECUTED in a container that has no Erlang/OTP.
 * Terms are tagged C structs owned by an environment; resources are reference-counted as in ERTS (alloc = 1 reference, a term made
 * of a resource holds one, the destructor runs when the last goes).  Written from the published erl_nif documentation; it pins
 * nothing about OTP itself -- what it shows is that the shim's own code runs, returns what libbwtgpu returned, and frees what it
 * allocated.  tests/nif_check/erl_nif.h declares the API the shim sees; this header is the test driver's side. */
#ifndef BWT_TEST_FAKE_ENIF_H
#define BWT_TEST_FAKE_ENIF_H
#include <stdio.h>
#include "erl_nif.h"
ErlNifEnv *fake_env_new(void);
void fake_env_free(ErlNifEnv *env);                 /* frees every term of the environment; releases the resources its terms hold */
ERL_NIF_TERM fake_make_binary(ErlNifEnv *env, const void *data, size_t n);
ERL_NIF_TERM fake_make_improper(ErlNifEnv *env, ERL_NIF_TERM head, ERL_NIF_TERM tail);      /* [H | T] with any T */
int fake_is_badarg(ERL_NIF_TERM t);
void fake_print(FILE *f, ERL_NIF_TERM t);           /* Erlang-like text; lists of bytes that are all printable print as "strings" */
long fake_live_terms(void);
long fake_live_resources(void);                     /* resource objects not yet destroyed */
long fake_dtor_calls(void);
#endif

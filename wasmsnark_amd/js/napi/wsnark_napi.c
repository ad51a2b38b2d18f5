/*
 * wsnark_napi.c -- thin N-API addon over the C ABI of libwsnark.so (include/wsnark.h).
 *
 * This is the binding a wasmsnark maintainer would add to replace the worker pool: each JS
 * function maps 1:1 to one C entry point and returns a Promise (napi async work, so the event
 * loop is never blocked), with the reference's byte layouts (Jacobian-Montgomery 96/192 B,
 * plain-form h, ...).  Reference seam: src/bn128.js:102-166 (worker commands), :353-415, :569-720.
 * libwsnark.so (../libwsnark.so next to this addon: the in-tree hipcc build) is dlopen'ed by init();
 * there is no fallback, no path argument and no environment override: if it cannot be loaded or finds no GPU,
 * init() throws.
 */
#define NAPI_VERSION 4
#define _GNU_SOURCE
#include <dlfcn.h>
#include <node_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct wsnark_pkey wsnark_pkey_t;
typedef struct wsnark_points wsnark_points_t;
typedef struct wsnark_group wsnark_group_t;
typedef struct wsnark_group_pkey wsnark_group_pkey_t;
static struct {
    void* h;
    int (*init)(int);
    void (*shutdown)(void);
    const char* (*last_error)(void);
    const char* (*device_info)(void);
    int (*g1_msm)(const void*, const void*, uint64_t, void*);
    int (*g2_msm)(const void*, const void*, uint64_t, void*);
    int (*fr_ntt)(void*, uint64_t, int, int);
    int (*calc_h)(const void*, const void*, size_t, const void*, size_t, uint32_t, uint32_t, void*);
    int (*pkey_load)(const void*, size_t, wsnark_pkey_t**);
    void (*pkey_free)(wsnark_pkey_t*);
    int (*pkey_info)(const wsnark_pkey_t*, uint32_t*, uint32_t*, uint32_t*);
    int (*prove)(wsnark_pkey_t*, const void*, size_t, const void*, const void*, void*);
    int (*last_blinding)(void*, void*);
    int (*verify)(const void*, size_t, const void*, uint64_t, const void*, int*);
    int (*host_alloc)(size_t, void**);
    void (*host_free)(void*);
    int (*pkey_load_stats)(const wsnark_pkey_t*, double*);
    int (*pkey_wait_tables)(wsnark_pkey_t*);
    int (*points_load)(int, const void*, uint64_t, wsnark_points_t**);
    void (*points_free)(wsnark_points_t*);
    int (*points_info)(const wsnark_points_t*, int*, uint64_t*, uint32_t*, uint32_t*, uint64_t*);
    int (*points_msm)(wsnark_points_t*, const void*, uint64_t, void*);
    /* several GPUs in this one process (include/wsnark.h: wsnark_group_*) */
    int (*group_create)(const int*, uint32_t, wsnark_group_t**);
    void (*group_free)(wsnark_group_t*);
    int (*group_pkey_load)(wsnark_group_t*, const void*, size_t, wsnark_group_pkey_t**);
    void (*group_pkey_free)(wsnark_group_pkey_t*);
    int (*group_pkey_info)(const wsnark_group_pkey_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, int*);
    int (*group_pkey_wait_tables)(wsnark_group_pkey_t*);
    int (*group_prove)(wsnark_group_pkey_t*, const void*, size_t, const void*, const void*, void*);
    int (*group_last_blinding)(wsnark_group_t*, void*, void*);
    int (*group_g1_msm)(wsnark_group_t*, const void*, const void*, uint64_t, void*);
    int (*group_g2_msm)(wsnark_group_t*, const void*, const void*, uint64_t, void*);
    /* key files: proving_key.bin or the WSNARK64 container (include/wsnark.h: wsnark_pkey_load_file) */
    int (*pkey_load_file)(const char*, uint32_t, uint32_t, uint32_t, wsnark_pkey_t**);
    int (*pkey_file_info)(const char*, uint32_t*, uint32_t*, uint32_t*, uint64_t*, int*);
    int (*group_pkey_load_file)(wsnark_group_t*, const char*, wsnark_group_pkey_t**);
    char dir[4096];
} L;

#define CHECK(env, call)                                                        \
    do {                                                                        \
        if ((call) != napi_ok) {                                                \
            napi_throw_error((env), NULL, "wsnark_napi: N-API call failed: " #call); \
            return NULL;                                                        \
        }                                                                       \
    } while (0)

/* The library this addon binds, relative to the addon's own directory: the in-tree hipcc build.  No argument and no environment
 * variable changes it; the test-suite's emulator host is a SEPARATE build of this file (make -C wasmsnark_amd/js emul ->
 * tests/emul/wsnark_napi_emul.node, compiled with another WSNARK_NAPI_LIB_RELPATH) that the product never loads. */
#ifndef WSNARK_NAPI_LIB_RELPATH
#define WSNARK_NAPI_LIB_RELPATH "../../libwsnark.so"
#endif
static int load_lib(char* err, size_t errlen) {
    char path[4200];
    if (L.h) return 0;
    snprintf(path, sizeof path, "%s/%s", L.dir, WSNARK_NAPI_LIB_RELPATH);
    L.h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!L.h) { snprintf(err, errlen, "cannot load %s: %s", path, dlerror()); return -1; }
#define SYM(field, name)                                                            \
    *(void**)(&L.field) = dlsym(L.h, name);                                         \
    if (!L.field) { snprintf(err, errlen, "%s lacks symbol %s", path, name); return -1; }
    SYM(init, "wsnark_init") SYM(shutdown, "wsnark_shutdown") SYM(last_error, "wsnark_last_error")
    SYM(device_info, "wsnark_device_info") SYM(g1_msm, "wsnark_g1_msm") SYM(g2_msm, "wsnark_g2_msm")
    SYM(fr_ntt, "wsnark_fr_ntt") SYM(calc_h, "wsnark_calc_h") SYM(pkey_load, "wsnark_pkey_load")
    SYM(pkey_free, "wsnark_pkey_free") SYM(pkey_info, "wsnark_pkey_info") SYM(prove, "wsnark_groth16_prove")
    SYM(last_blinding, "wsnark_last_blinding") SYM(verify, "wsnark_groth16_verify")
    SYM(host_alloc, "wsnark_host_alloc") SYM(host_free, "wsnark_host_free")
    SYM(pkey_load_stats, "wsnark_pkey_load_stats") SYM(pkey_wait_tables, "wsnark_pkey_wait_tables")
    SYM(points_load, "wsnark_points_load") SYM(points_free, "wsnark_points_free") SYM(points_info, "wsnark_points_info") SYM(points_msm, "wsnark_points_msm")
    SYM(group_create, "wsnark_group_create") SYM(group_free, "wsnark_group_free") SYM(group_pkey_load, "wsnark_group_pkey_load")
    SYM(group_pkey_free, "wsnark_group_pkey_free") SYM(group_pkey_info, "wsnark_group_pkey_info")
    SYM(group_pkey_wait_tables, "wsnark_group_pkey_wait_tables") SYM(group_prove, "wsnark_group_prove")
    SYM(group_last_blinding, "wsnark_group_last_blinding") SYM(group_g1_msm, "wsnark_group_g1_msm") SYM(group_g2_msm, "wsnark_group_g2_msm")
    SYM(pkey_load_file, "wsnark_pkey_load_file") SYM(pkey_file_info, "wsnark_pkey_file_info") SYM(group_pkey_load_file, "wsnark_group_pkey_load_file")
#undef SYM
    return 0;
}

/* ArrayBuffer | Buffer | TypedArray | DataView -> (ptr, len); returns 0 on failure */
static int get_bytes(napi_env env, napi_value v, uint8_t** p, size_t* n) {
    bool is;
    if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) return napi_get_arraybuffer_info(env, v, (void**)p, n) == napi_ok;
    if (napi_is_buffer(env, v, &is) == napi_ok && is) return napi_get_buffer_info(env, v, (void**)p, n) == napi_ok;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type t; size_t len, off; napi_value ab; void* data;
        if (napi_get_typedarray_info(env, v, &t, &len, &data, &ab, &off) != napi_ok) return 0;
        static const size_t esz[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
        *p = (uint8_t*)data; *n = len * esz[t];
        return 1;
    }
    if (napi_is_dataview(env, v, &is) == napi_ok && is) {
        size_t len, off; napi_value ab; void* data;
        if (napi_get_dataview_info(env, v, &len, &data, &ab, &off) != napi_ok) return 0;
        *p = (uint8_t*)data; *n = len;
        return 1;
    }
    return 0;
}

enum { OP_G1, OP_G2, OP_NTT, OP_CALCH, OP_PROVE, OP_LOADKEY, OP_VERIFY, OP_HASH, OP_WAIT_TABLES,
       OP_GROUP_G1, OP_GROUP_G2, OP_GROUP_LOADKEY, OP_GROUP_PROVE, OP_GROUP_WAIT_TABLES, OP_POINTS_LOAD, OP_POINTS_MSM,
       OP_LOADKEY_FILE, OP_GROUP_LOADKEY_FILE };
/* A group and the keys loaded on it.  The JS side holds them as externals; a key's finalizer must not touch a group that
 * terminate() has already freed (wsnark_group_free frees the keys that are left), so every group handle carries a `live` flag
 * that outlives the group itself and every key handle points at its group's handle. */
/* Lifetime: jobs queued on the libuv pool hold raw pointers into the group, so the library's group must outlive every
 * job that was queued while it was live.  `inflight` counts them (touched on the JS thread only: start_job / job_complete);
 * groupFree / terminate() / the finalizer only mark the group dead (`live` = 0, read by the pool threads) and the LAST job's
 * completion frees it -- at once when nothing is in flight.  A job that reaches the pool after the group died is rejected
 * without entering the library. */
/* Every external this addon hands to JS starts with a tag naming its kind (Node 12 has no napi_type_tag_object): a group key passed
 * where a key handle is expected -- index.js once did that after terminate() -- is refused instead of being read as the other struct. */
#define TAG_KEY 0x77736e4b45593031ull
#define TAG_POINTS 0x77736e5054533031ull
#define TAG_GROUP 0x77736e4752503031ull
#define TAG_GKEY 0x77736e474b593031ull
typedef struct { uint64_t tag; void* p; } handle_t;                 /* a key (wsnark_pkey_t*) or a point set (wsnark_points_t*) */
typedef struct { uint64_t tag; wsnark_group_t* g; int live; int inflight; int refs; } group_ref_t;
typedef struct { uint64_t tag; wsnark_group_pkey_t* k; group_ref_t* gr; } gkey_ref_t;
static void* get_handle(napi_env env, napi_value v, uint64_t tag) {
    void* p = NULL;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_external || napi_get_value_external(env, v, &p) != napi_ok || !p) return NULL;
    return *(const uint64_t*)p == tag ? p : NULL;
}
static void* get_plain(napi_env env, napi_value v, uint64_t tag) {   /* the library pointer inside a key / points handle */
    handle_t* h = (handle_t*)get_handle(env, v, tag);
    return h ? h->p : NULL;
}
static handle_t* new_handle(uint64_t tag, void* p) {
    handle_t* h = (handle_t*)calloc(1, sizeof *h);
    if (h) { h->tag = tag; h->p = p; }
    return h;
}
static void group_ref_drop(group_ref_t* gr) { if (--gr->refs == 0) { gr->tag = 0; free(gr); } }
static void group_release(group_ref_t* gr);
typedef struct {
    int op, rc;
    napi_async_work work;
    napi_deferred deferred;
    napi_ref refs[4];
    int nrefs;
    uint8_t *a, *b, *c, *r32, *s32;
    size_t na, nb, nc;
    uint32_t u0, u1;
    int i0, i1;
    wsnark_pkey_t* key;
    group_ref_t* gr;
    gkey_ref_t* gk;
    group_ref_t* gr_used;       /* the group this job runs on (counted in its `inflight`), or NULL */
    wsnark_points_t* pts;
    uint8_t* out;
    size_t nout;
    char* path;                 /* key file (OP_LOADKEY_FILE, OP_GROUP_LOADKEY_FILE): owned by the job */
    char err[512];
} job_t;

/* 128-bit digest of a run of bytes: four independent multiply-rotate lanes over 8-byte words, folded with the length.
 * Not cryptographic: it only has to notice that the bytes behind a cached key handle are no longer the bytes that were loaded. */
static void hash_run(const uint8_t* p, size_t n, uint8_t out[16]) {
    uint64_t h[4] = {0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull, 0x27d4eb2f165667c5ull};
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        for (int k = 0; k < 4; k++) {
            h[k] = (h[k] ^ w[k]) * 0xff51afd7ed558ccdull;
            h[k] = (h[k] << 29) | (h[k] >> 35);
        }
    }
    uint64_t tail[4] = {0, 0, 0, 0};
    memcpy(tail, p + i, n - i);
    for (int k = 0; k < 4; k++) h[k] = (h[k] ^ tail[k]) * 0xc4ceb9fe1a85ec53ull;
    uint64_t a = h[0] ^ ((h[1] << 21) | (h[1] >> 43)) ^ (uint64_t)n, b = h[2] ^ ((h[3] << 37) | (h[3] >> 27));
    a = (a ^ (a >> 33)) * 0xff51afd7ed558ccdull; b = (b ^ (b >> 29)) * 0xc4ceb9fe1a85ec53ull;
    a ^= b >> 31; b ^= a >> 27;
    memcpy(out, &a, 8); memcpy(out + 8, &b, 8);
}
/* ... of a whole buffer: the digests of its 4 MiB blocks, digested.  One core streams about 12 GB/s through hash_run -- 50 ms
 * for a 0.6 GB key, longer than the key's load takes -- so the blocks are dealt to up to HASH_THREADS threads (the result does
 * not depend on how many there were). */
#define HASH_BLOCK ((size_t)4 << 20)
#define HASH_THREADS 8
typedef struct { const uint8_t* p; size_t n, nblocks; uint8_t* digests; unsigned first, step; } hash_part_t;
static void* hash_part(void* arg) {
    hash_part_t* h = (hash_part_t*)arg;
    for (size_t b = h->first; b < h->nblocks; b += h->step) {
        const size_t lo = b * HASH_BLOCK, len = h->n - lo < HASH_BLOCK ? h->n - lo : HASH_BLOCK;
        hash_run(h->p + lo, len, h->digests + 16 * b);
    }
    return NULL;
}
static int hash_bytes(const uint8_t* p, size_t n, uint8_t out[16]) {
    const size_t nblocks = n ? (n + HASH_BLOCK - 1) / HASH_BLOCK : 1;
    uint8_t* digests = (uint8_t*)calloc(nblocks, 16);
    if (!digests) return -1;
    unsigned want = nblocks < HASH_THREADS ? (unsigned)nblocks : HASH_THREADS;
    hash_part_t parts[HASH_THREADS];
    pthread_t tid[HASH_THREADS];
    unsigned started = 0;
    for (unsigned t = 0; t < want; t++) parts[t] = (hash_part_t){p, n, nblocks, digests, t, want};
    for (unsigned t = 1; t < want; t++) {
        if (pthread_create(&tid[t], NULL, hash_part, &parts[t]) != 0) break;
        started = t;
    }
    /* (threads that could not be started: their blocks are done here, after this thread's own) */
    hash_part(&parts[0]);
    for (unsigned t = started + 1; t < want; t++) hash_part(&parts[t]);
    for (unsigned t = 1; t <= started; t++) pthread_join(tid[t], NULL);
    uint8_t top[16];
    hash_run(digests, nblocks * 16, top);
    uint64_t a, b;
    memcpy(&a, top, 8); memcpy(&b, top + 8, 8);
    a ^= (uint64_t)n * 0x9e3779b97f4a7c15ull;
    memcpy(out, &a, 8); memcpy(out + 8, &b, 8);
    free(digests);
    return 0;
}

static void job_execute(napi_env env, void* data) {
    (void)env;
    job_t* j = (job_t*)data;
    if (j->gr_used && !__atomic_load_n(&j->gr_used->live, __ATOMIC_ACQUIRE)) {      /* terminate() ran after this job was queued */
        j->rc = -1;
        snprintf(j->err, sizeof j->err, "wsnark: the group was terminated before this call ran");
        return;
    }
    switch (j->op) {
    case OP_G1: j->rc = L.g1_msm(j->a, j->b, j->na / 32, j->out); break;
    case OP_G2: j->rc = L.g2_msm(j->a, j->b, j->na / 32, j->out); break;
    case OP_NTT: memcpy(j->out, j->a, j->na); j->rc = L.fr_ntt(j->out, j->na / 32, j->i0, j->i1); break;
    case OP_CALCH: j->rc = L.calc_h(j->a, j->b, j->nb, j->c, j->nc, j->u0, j->u1, j->out); break;
    case OP_PROVE:   /* out = proof (384 B) | r | s actually used (same pool thread: wsnark_last_blinding is per thread) */
        j->rc = L.prove(j->key, j->a, j->na, j->r32, j->s32, j->out);
        if (!j->rc) j->rc = L.last_blinding(j->out + 384, j->out + 416);
        break;
    case OP_LOADKEY: j->rc = L.pkey_load(j->a, j->na, &j->key); break;
    case OP_LOADKEY_FILE: j->rc = L.pkey_load_file(j->path, 0, 1, 0, &j->key); break;
    case OP_GROUP_LOADKEY_FILE: j->rc = L.group_pkey_load_file(j->gr->g, j->path, &j->gk->k); break;
    case OP_VERIFY: j->rc = L.verify(j->a, j->na, j->b, j->nb / 32, j->c, &j->i0); break;
    case OP_WAIT_TABLES: j->rc = L.pkey_wait_tables(j->key); break;
    case OP_GROUP_G1: j->rc = L.group_g1_msm(j->gr->g, j->a, j->b, j->na / 32, j->out); break;
    case OP_GROUP_G2: j->rc = L.group_g2_msm(j->gr->g, j->a, j->b, j->na / 32, j->out); break;
    case OP_GROUP_LOADKEY: j->rc = L.group_pkey_load(j->gr->g, j->a, j->na, &j->gk->k); break;
    case OP_GROUP_PROVE:
        j->rc = L.group_prove(j->gk->k, j->a, j->na, j->r32, j->s32, j->out);
        if (!j->rc) j->rc = L.group_last_blinding(j->gk->gr->g, j->out + 384, j->out + 416);
        break;
    case OP_GROUP_WAIT_TABLES: j->rc = L.group_pkey_wait_tables(j->gk->k); break;
    case OP_POINTS_LOAD: j->rc = L.points_load(j->i0, j->a, j->na / (j->i0 == 2 ? 128 : 64), &j->pts); break;
    case OP_POINTS_MSM: j->rc = L.points_msm(j->pts, j->a, j->na / 32, j->out); break;
    case OP_HASH:
        if (hash_bytes(j->a, j->na, j->out)) { j->rc = -1; snprintf(j->err, sizeof j->err, "hashBytes: out of memory"); return; }
        break;
    }
    if (j->rc) snprintf(j->err, sizeof j->err, "wsnark error %d: %s", j->rc, L.last_error());
}

/* JS thread.  The group dies now if no job is in flight on it, otherwise with the last of them (job_complete). */
static void group_release(group_ref_t* gr) {
    __atomic_store_n(&gr->live, 0, __ATOMIC_RELEASE);
    if (gr->g && gr->inflight == 0) { L.group_free(gr->g); gr->g = NULL; }
}
static void job_use_group(job_t* j, group_ref_t* gr) { j->gr_used = gr; gr->inflight++; gr->refs++; }
static void job_done_with_group(job_t* j) {
    group_ref_t* gr = j->gr_used;
    if (!gr) return;
    j->gr_used = NULL;
    gr->inflight--;
    if (!gr->live && gr->inflight == 0 && gr->g) { L.group_free(gr->g); gr->g = NULL; }
    group_ref_drop(gr);
}

static void key_finalize(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    handle_t* h = (handle_t*)data;
    if (h) { if (h->p) L.pkey_free((wsnark_pkey_t*)h->p); h->tag = 0; free(h); }
}

static void points_finalize(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    handle_t* h = (handle_t*)data;
    if (h) { if (h->p) L.points_free((wsnark_points_t*)h->p); h->tag = 0; free(h); }
}
static void gkey_finalize(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    gkey_ref_t* gk = (gkey_ref_t*)data;
    if (!gk) return;
    if (gk->k && gk->gr->live) L.group_pkey_free(gk->k);      /* (a dead or dying group takes its keys with it) */
    group_ref_drop(gk->gr);
    gk->tag = 0;
    free(gk);
}
static void group_finalize(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    group_ref_t* gr = (group_ref_t*)data;
    if (!gr) return;
    group_release(gr);          /* (jobs in flight keep `gr` alive through their own reference and free the group when they finish) */
    group_ref_drop(gr);
}

static void job_complete(napi_env env, napi_status status, void* data) {
    job_t* j = (job_t*)data;
    napi_value res;
    if (status != napi_ok || j->rc) {
        napi_value msg, e;
        napi_create_string_utf8(env, j->rc ? j->err : "async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &e);
        napi_reject_deferred(env, j->deferred, e);
        if ((j->op == OP_GROUP_LOADKEY || j->op == OP_GROUP_LOADKEY_FILE) && j->gk) { group_ref_drop(j->gk->gr); free(j->gk); }
    } else if (j->op == OP_GROUP_LOADKEY || j->op == OP_GROUP_LOADKEY_FILE) {
        napi_create_external(env, j->gk, gkey_finalize, NULL, &res);
        napi_resolve_deferred(env, j->deferred, res);
    } else if (j->op == OP_POINTS_LOAD) {
        napi_create_external(env, new_handle(TAG_POINTS, j->pts), points_finalize, NULL, &res);
        napi_resolve_deferred(env, j->deferred, res);
    } else if (j->op == OP_VERIFY) {
        napi_get_boolean(env, j->i0 != 0, &res);
        napi_resolve_deferred(env, j->deferred, res);
    } else if (j->op == OP_LOADKEY || j->op == OP_LOADKEY_FILE) {
        napi_create_external(env, new_handle(TAG_KEY, j->key), key_finalize, NULL, &res);
        napi_resolve_deferred(env, j->deferred, res);
    } else {
        void* dst;
        napi_create_arraybuffer(env, j->nout, &dst, &res);   /* results are fresh ArrayBuffers, as in the reference */
        memcpy(dst, j->out, j->nout);
        napi_resolve_deferred(env, j->deferred, res);
    }
    for (int i = 0; i < j->nrefs; i++) napi_delete_reference(env, j->refs[i]);
    napi_delete_async_work(env, j->work);
    job_done_with_group(j);
    free(j->path);
    free(j->out);
    free(j);
}

static napi_value start_job(napi_env env, job_t* j, const char* name) {
    napi_value promise, rname;
    if (!L.h) { job_done_with_group(j); free(j->path); free(j->out); free(j); napi_throw_error(env, NULL, "wsnark_napi: init() has not been called (use buildBn128())"); return NULL; }
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok || napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &rname) != napi_ok ||
        napi_create_async_work(env, NULL, rname, job_execute, job_complete, j, &j->work) != napi_ok || napi_queue_async_work(env, j->work) != napi_ok) {
        for (int i = 0; i < j->nrefs; i++) napi_delete_reference(env, j->refs[i]);
        job_done_with_group(j);           /* (never queued: nothing will complete it) */
        free(j->path); free(j->out); free(j);
        napi_throw_error(env, NULL, "wsnark_napi: cannot queue the call");
        return NULL;
    }
    return promise;
}
static int keep(napi_env env, job_t* j, napi_value v) {   /* inputs stay referenced until completion */
    return napi_create_reference(env, v, 1, &j->refs[j->nrefs++]) == napi_ok;
}
#define FAIL(env, j, msg) do { free((j)->path); free((j)->out); free(j); napi_throw_type_error((env), NULL, (msg)); return NULL; } while (0)
static char* get_path(napi_env env, napi_value v) {        /* a JS string -> malloc'ed UTF-8 (NULL if it is not a string) */
    size_t n = 0;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_string || napi_get_value_string_utf8(env, v, NULL, 0, &n) != napi_ok) return NULL;
    char* p = (char*)malloc(n + 1);
    if (p && napi_get_value_string_utf8(env, v, p, n + 1, &n) != napi_ok) { free(p); return NULL; }
    return p;
}

/* g1Multiexp(scalars, points) / g2Multiexp(scalars, points) -> Promise<ArrayBuffer 96/192> */
static napi_value msm_common(napi_env env, napi_callback_info info, int op) {
    size_t argc = 2; napi_value argv[2];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = op; j->nout = op == OP_G1 ? 96 : 192; j->out = (uint8_t*)malloc(j->nout);
    if (argc < 2 || !get_bytes(env, argv[0], &j->a, &j->na) || !get_bytes(env, argv[1], &j->b, &j->nb))
        FAIL(env, j, "expected (scalars, points) byte buffers");
    if (j->nb < (j->na / 32) * (op == OP_G1 ? 64 : 128)) FAIL(env, j, "points buffer too short for the number of scalars");
    keep(env, j, argv[0]); keep(env, j, argv[1]);
    return start_job(env, j, op == OP_G1 ? "wsnark_g1_msm" : "wsnark_g2_msm");
}
static napi_value js_g1(napi_env env, napi_callback_info info) { return msm_common(env, info, OP_G1); }
static napi_value js_g2(napi_env env, napi_callback_info info) { return msm_common(env, info, OP_G2); }

/* fft(buf, odd, inverse) -> Promise<ArrayBuffer> (out of place: the input is left untouched) */
static napi_value js_fft(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_NTT;
    if (argc < 1 || !get_bytes(env, argv[0], &j->a, &j->na)) FAIL(env, j, "expected a byte buffer");
    if (argc > 1) napi_get_value_int32(env, argv[1], &j->i0);
    if (argc > 2) { bool b = false; napi_coerce_to_bool(env, argv[2], &argv[2]); napi_get_value_bool(env, argv[2], &b); j->i1 = b; }
    j->nout = j->na; j->out = (uint8_t*)malloc(j->na ? j->na : 1);
    keep(env, j, argv[0]);
    return start_job(env, j, "wsnark_fr_ntt");
}

/* calcH(signals, polsA, polsB, nSignals, domainSize) -> Promise<ArrayBuffer domain*32> */
static napi_value js_calch(napi_env env, napi_callback_info info) {
    size_t argc = 5; napi_value argv[5];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_CALCH;
    if (argc < 5 || !get_bytes(env, argv[0], &j->a, &j->na) || !get_bytes(env, argv[1], &j->b, &j->nb) ||
        !get_bytes(env, argv[2], &j->c, &j->nc) || napi_get_value_uint32(env, argv[3], &j->u0) != napi_ok ||
        napi_get_value_uint32(env, argv[4], &j->u1) != napi_ok)
        FAIL(env, j, "expected (signals, polsA, polsB, nSignals, domainSize)");
    if (j->na < (size_t)j->u0 * 32) FAIL(env, j, "signals shorter than nSignals*32 bytes");
    j->nout = (size_t)j->u1 * 32; j->out = (uint8_t*)malloc(j->nout ? j->nout : 1);
    keep(env, j, argv[0]); keep(env, j, argv[1]); keep(env, j, argv[2]);
    return start_job(env, j, "wsnark_calc_h");
}

/* loadKey(pkey) -> Promise<external handle>; freed by the GC finalizer */
static napi_value js_loadkey(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_LOADKEY;
    if (argc < 1 || !get_bytes(env, argv[0], &j->a, &j->na)) FAIL(env, j, "expected a proving_key.bin byte buffer");
    keep(env, j, argv[0]);
    return start_job(env, j, "wsnark_pkey_load");
}

/* loadKeyFile(path) -> Promise<key handle>: proving_key.bin or the WSNARK64 container, mapped by the library (wsnark_pkey_load_file):
 * the key never exists as a JS buffer -- how keys beyond one ArrayBuffer (4 GiB file format, 2^24 constraints = 7.8 GB) are loaded */
static napi_value js_loadkey_file(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_LOADKEY_FILE;
    if (argc < 1 || !(j->path = get_path(env, argv[0]))) FAIL(env, j, "expected the path of a key file");
    return start_job(env, j, "wsnark_pkey_load_file");
}
/* keyFileInfo(path) -> {nVars, nPublic, domainSize, fileBytes, format: "proving_key.bin" | "WSNARK64"} (synchronous: reads the header) */
static napi_value js_keyfile_info(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1], o, v;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!L.h) { napi_throw_error(env, NULL, "wsnark_napi: init() has not been called (use buildBn128())"); return NULL; }
    char* path = argc >= 1 ? get_path(env, argv[0]) : NULL;
    if (!path) { napi_throw_type_error(env, NULL, "expected the path of a key file"); return NULL; }
    uint32_t nv = 0, np = 0, dom = 0; uint64_t nb = 0; int fmt = 0;
    int rc = L.pkey_file_info(path, &nv, &np, &dom, &nb, &fmt);
    free(path);
    if (rc) {
        char msg[600];
        snprintf(msg, sizeof msg, "wsnark error %d: %s", rc, L.last_error());
        napi_throw_error(env, NULL, msg);
        return NULL;
    }
    napi_create_object(env, &o);
    napi_create_uint32(env, nv, &v); napi_set_named_property(env, o, "nVars", v);
    napi_create_uint32(env, np, &v); napi_set_named_property(env, o, "nPublic", v);
    napi_create_uint32(env, dom, &v); napi_set_named_property(env, o, "domainSize", v);
    napi_create_double(env, (double)nb, &v); napi_set_named_property(env, o, "fileBytes", v);
    napi_create_string_utf8(env, fmt == 2 ? "WSNARK64" : "proving_key.bin", NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, o, "format", v);
    return o;
}

/* hashBytes(buf) -> Promise<ArrayBuffer 16>: digest of the WHOLE buffer, computed off the event loop */
static napi_value js_hash(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_HASH; j->nout = 16; j->out = (uint8_t*)malloc(16);
    if (argc < 1 || !get_bytes(env, argv[0], &j->a, &j->na)) FAIL(env, j, "expected a byte buffer");
    keep(env, j, argv[0]);
    return start_job(env, j, "wsnark_hash_bytes");
}

/* waitTables(keyHandle) -> Promise: resolves once the key's fixed-base table rows are built (wsnark_pkey_wait_tables; proofs before
 * that run on the plain sections) */
static napi_value js_wait_tables(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_WAIT_TABLES; j->nout = 1; j->out = (uint8_t*)calloc(1, 1);
    if (argc < 1 || !(j->key = (wsnark_pkey_t*)get_plain(env, argv[0], TAG_KEY))) FAIL(env, j, "expected a key handle");
    keep(env, j, argv[0]);
    return start_job(env, j, "wsnark_pkey_wait_tables");
}

/* prove(keyHandle, witness, r32|null, s32|null) -> Promise<ArrayBuffer 448>: proof (384 B) | r | s used */
static napi_value js_prove(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value argv[4];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_PROVE; j->nout = 448; j->out = (uint8_t*)malloc(448);
    if (argc < 2 || !(j->key = (wsnark_pkey_t*)get_plain(env, argv[0], TAG_KEY)) ||
        !get_bytes(env, argv[1], &j->a, &j->na))
        FAIL(env, j, "expected (keyHandle, witness[, r32, s32])");
    keep(env, j, argv[0]); keep(env, j, argv[1]);
    size_t n;
    napi_valuetype t;
    if (argc > 2 && napi_typeof(env, argv[2], &t) == napi_ok && t != napi_null && t != napi_undefined) {
        if (!get_bytes(env, argv[2], &j->r32, &n) || n != 32) FAIL(env, j, "r must be 32 bytes");
        keep(env, j, argv[2]);
    }
    if (argc > 3 && napi_typeof(env, argv[3], &t) == napi_ok && t != napi_null && t != napi_undefined) {
        if (!get_bytes(env, argv[3], &j->s32, &n) || n != 32) FAIL(env, j, "s must be 32 bytes");
        keep(env, j, argv[3]);
    }
    return start_job(env, j, "wsnark_groth16_prove");
}

/* verify(vkBytes, inputBytes, proof384) -> Promise<boolean>   (layouts: include/wsnark.h, wsnark_groth16_verify) */
static napi_value js_verify(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_VERIFY;
    if (argc < 3 || !get_bytes(env, argv[0], &j->a, &j->na) || !get_bytes(env, argv[1], &j->b, &j->nb) ||
        !get_bytes(env, argv[2], &j->c, &j->nc) || j->nc != 384 || j->nb % 32)
        FAIL(env, j, "expected (vkBytes, inputBytes (n x 32), proof384)");
    keep(env, j, argv[0]); keep(env, j, argv[1]); keep(env, j, argv[2]);
    return start_job(env, j, "wsnark_groth16_verify");
}

/* allocPinned(bytes) -> ArrayBuffer over pinned host memory (wsnark_host_alloc): a witness written into it is DMA'd in place,
 * without the staging copy.  Freed by the GC finalizer. */
static void pinned_finalize(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    if (data && L.host_free) L.host_free(data);
}
static napi_value js_alloc_pinned(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1], ab;
    double nbytes = 0;
    void* p = NULL;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!L.h) { napi_throw_error(env, NULL, "wsnark_napi: init() has not been called (use buildBn128())"); return NULL; }
    if (argc < 1 || napi_get_value_double(env, argv[0], &nbytes) != napi_ok || nbytes < 0 || nbytes > 1e12) { napi_throw_type_error(env, NULL, "expected a byte count"); return NULL; }
    int rc = L.host_alloc((size_t)nbytes, &p);
    if (rc || !p) {
        char msg[600];
        snprintf(msg, sizeof msg, "wsnark_host_alloc failed (%d): %s", rc, L.last_error());
        napi_throw_error(env, NULL, msg);
        return NULL;
    }
    if (napi_create_external_arraybuffer(env, p, (size_t)nbytes, pinned_finalize, NULL, &ab) != napi_ok) {
        L.host_free(p);
        napi_throw_error(env, NULL, "wsnark_napi: cannot wrap the pinned buffer");
        return NULL;
    }
    return ab;
}

static napi_value js_keyinfo(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1], o, v;
    wsnark_pkey_t* k = NULL;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 1 || !(k = (wsnark_pkey_t*)get_plain(env, argv[0], TAG_KEY))) { napi_throw_type_error(env, NULL, "expected a key handle"); return NULL; }
    uint32_t nv, np, dom;
    L.pkey_info(k, &nv, &np, &dom);
    napi_create_object(env, &o);
    napi_create_uint32(env, nv, &v); napi_set_named_property(env, o, "nVars", v);
    napi_create_uint32(env, np, &v); napi_set_named_property(env, o, "nPublic", v);
    napi_create_uint32(env, dom, &v); napi_set_named_property(env, o, "domainSize", v);
    /* what the load took, phase by phase (wsnark_pkey_load_stats; tableBuild stays 0 until the background build has finished) */
    double ms[5] = {0, 0, 0, 0, 0};
    if (L.pkey_load_stats(k, ms) == 0) {
        static const char* names[5] = {"polsToCsr", "pointsH2d", "masksConvert", "tableBuild", "total"};
        napi_value lo;
        napi_create_object(env, &lo);
        for (int i = 0; i < 5; i++) { napi_create_double(env, ms[i], &v); napi_set_named_property(env, lo, names[i], v); }
        napi_set_named_property(env, o, "loadMs", lo);
    }
    return o;
}

/* ---- resident bases ----
 * loadPoints(group 1|2, points) -> Promise<handle>: the set resident as fixed-base window tables (wsnark_points_load) */
static napi_value js_points_load(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_POINTS_LOAD;
    if (argc < 2 || napi_get_value_int32(env, argv[0], &j->i0) != napi_ok || (j->i0 != 1 && j->i0 != 2) || !get_bytes(env, argv[1], &j->a, &j->na) ||
        j->na == 0 || j->na % (j->i0 == 2 ? 128 : 64))
        FAIL(env, j, "expected (1 | 2, a whole number of affine points)");
    keep(env, j, argv[1]);
    return start_job(env, j, "wsnark_points_load");
}
/* pointsMultiexp(handle, scalars) -> Promise<ArrayBuffer 96/192> */
static napi_value js_points_msm(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    int g = 0;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_POINTS_MSM;
    if (argc < 2 || !(j->pts = (wsnark_points_t*)get_plain(env, argv[0], TAG_POINTS)) || !get_bytes(env, argv[1], &j->a, &j->na))
        FAIL(env, j, "expected (points handle, scalars)");
    L.points_info(j->pts, &g, NULL, NULL, NULL, NULL);
    j->nout = g == 2 ? 192 : 96; j->out = (uint8_t*)malloc(j->nout);
    keep(env, j, argv[0]); keep(env, j, argv[1]);
    return start_job(env, j, "wsnark_points_msm");
}

/* ---- several GPUs in this one process ----
 * groupCreate([device, ...]) -> group handle (synchronous: contexts and worker threads are created at once) */
static napi_value js_group_create(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1], res;
    uint32_t n = 0;
    bool is_arr = false;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!L.h) { napi_throw_error(env, NULL, "wsnark_napi: init() has not been called (use buildBn128())"); return NULL; }
    if (argc < 1 || napi_is_array(env, argv[0], &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, argv[0], &n) != napi_ok || n == 0 || n > 64) {
        napi_throw_type_error(env, NULL, "expected a non-empty array of device ordinals");
        return NULL;
    }
    int devs[64];
    for (uint32_t i = 0; i < n; i++) {
        napi_value e; int32_t d = 0;
        if (napi_get_element(env, argv[0], i, &e) != napi_ok || napi_get_value_int32(env, e, &d) != napi_ok || d < 0) { napi_throw_type_error(env, NULL, "device ordinals must be non-negative integers"); return NULL; }
        devs[i] = d;
    }
    group_ref_t* gr = (group_ref_t*)calloc(1, sizeof *gr);
    int rc = L.group_create(devs, n, &gr->g);
    if (rc) {
        char msg[600];
        snprintf(msg, sizeof msg, "wsnark_group_create failed (%d): %s -- there is no CPU fallback", rc, L.last_error());
        free(gr);
        napi_throw_error(env, NULL, msg);
        return NULL;
    }
    gr->tag = TAG_GROUP; gr->live = 1; gr->refs = 1;
    if (napi_create_external(env, gr, group_finalize, NULL, &res) != napi_ok) { L.group_free(gr->g); free(gr); napi_throw_error(env, NULL, "wsnark_napi: cannot wrap the group"); return NULL; }
    return res;
}
/* groupFree(group): the group is dead from here on (new calls are refused, queued ones are rejected when they reach the pool);
 * its contexts and the keys still loaded on them are freed now, or by the completion of the last call in flight.  Never blocks
 * the event loop behind a running proof.  The handle stays valid as a dead one. */
static napi_value js_group_free(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1], u;
    group_ref_t* gr = NULL;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc >= 1 && (gr = (group_ref_t*)get_handle(env, argv[0], TAG_GROUP)) && gr->live) group_release(gr);
    napi_get_undefined(env, &u);
    return u;
}
static group_ref_t* live_group(napi_env env, napi_value v) {
    group_ref_t* gr = (group_ref_t*)get_handle(env, v, TAG_GROUP);
    return gr && gr->live ? gr : NULL;
}
static gkey_ref_t* live_gkey(napi_env env, napi_value v) {
    gkey_ref_t* gk = (gkey_ref_t*)get_handle(env, v, TAG_GKEY);
    return gk && gk->k && gk->gr->live ? gk : NULL;
}
/* groupMultiexp(group, which, scalars, points) -> Promise<ArrayBuffer 96/192> */
static napi_value js_group_msm(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value argv[4];
    int32_t which = 0;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    if (argc < 4 || !(j->gr = live_group(env, argv[0])) || napi_get_value_int32(env, argv[1], &which) != napi_ok ||
        !get_bytes(env, argv[2], &j->a, &j->na) || !get_bytes(env, argv[3], &j->b, &j->nb))
        FAIL(env, j, "expected (group, which, scalars, points)");
    j->op = which ? OP_GROUP_G2 : OP_GROUP_G1; j->nout = which ? 192 : 96; j->out = (uint8_t*)malloc(j->nout);
    if (j->nb < (j->na / 32) * (which ? 128 : 64)) FAIL(env, j, "points buffer too short for the number of scalars");
    keep(env, j, argv[0]); keep(env, j, argv[2]); keep(env, j, argv[3]);
    job_use_group(j, j->gr);
    return start_job(env, j, "wsnark_group_msm");
}
/* groupLoadKey(group, pkey) -> Promise<group key handle> */
static napi_value js_group_loadkey(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_GROUP_LOADKEY;
    if (argc < 2 || !(j->gr = live_group(env, argv[0])) || !get_bytes(env, argv[1], &j->a, &j->na)) FAIL(env, j, "expected (group, proving_key.bin bytes)");
    j->gk = (gkey_ref_t*)calloc(1, sizeof *j->gk);
    j->gk->tag = TAG_GKEY; j->gk->gr = j->gr; j->gr->refs++;
    keep(env, j, argv[0]); keep(env, j, argv[1]);
    job_use_group(j, j->gr);
    return start_job(env, j, "wsnark_group_pkey_load");
}
/* groupLoadKeyFile(group, path) -> Promise<group key handle>: every member reads its own shard's pages of ONE mapping */
static napi_value js_group_loadkey_file(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_GROUP_LOADKEY_FILE;
    if (argc < 2 || !(j->gr = live_group(env, argv[0])) || !(j->path = get_path(env, argv[1]))) FAIL(env, j, "expected (group, path of a key file)");
    j->gk = (gkey_ref_t*)calloc(1, sizeof *j->gk);
    j->gk->tag = TAG_GKEY; j->gk->gr = j->gr; j->gr->refs++;
    keep(env, j, argv[0]);
    job_use_group(j, j->gr);
    return start_job(env, j, "wsnark_group_pkey_load_file");
}
/* groupProve(groupKey, witness, r32|null, s32|null) -> Promise<ArrayBuffer 448>: proof | r | s used */
static napi_value js_group_prove(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value argv[4];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_GROUP_PROVE; j->nout = 448; j->out = (uint8_t*)malloc(448);
    if (argc < 2 || !(j->gk = live_gkey(env, argv[0])) || !get_bytes(env, argv[1], &j->a, &j->na)) FAIL(env, j, "expected (groupKey, witness[, r32, s32])");
    keep(env, j, argv[0]); keep(env, j, argv[1]);
    size_t n;
    napi_valuetype t;
    if (argc > 2 && napi_typeof(env, argv[2], &t) == napi_ok && t != napi_null && t != napi_undefined) {
        if (!get_bytes(env, argv[2], &j->r32, &n) || n != 32) FAIL(env, j, "r must be 32 bytes");
        keep(env, j, argv[2]);
    }
    if (argc > 3 && napi_typeof(env, argv[3], &t) == napi_ok && t != napi_null && t != napi_undefined) {
        if (!get_bytes(env, argv[3], &j->s32, &n) || n != 32) FAIL(env, j, "s must be 32 bytes");
        keep(env, j, argv[3]);
    }
    job_use_group(j, j->gk->gr);
    return start_job(env, j, "wsnark_group_prove");
}
static napi_value js_group_wait_tables(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    job_t* j = (job_t*)calloc(1, sizeof *j);
    j->op = OP_GROUP_WAIT_TABLES; j->nout = 1; j->out = (uint8_t*)calloc(1, 1);
    if (argc < 1 || !(j->gk = live_gkey(env, argv[0]))) FAIL(env, j, "expected a group key handle");
    keep(env, j, argv[0]);
    job_use_group(j, j->gk->gr);
    return start_job(env, j, "wsnark_group_pkey_wait_tables");
}
/* groupKeyInfo(groupKey) -> {nVars, nPublic, domainSize, world, distributedCalcH} */
static napi_value js_group_keyinfo(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1], o, v;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    gkey_ref_t* gk = argc >= 1 ? live_gkey(env, argv[0]) : NULL;
    if (!gk) { napi_throw_type_error(env, NULL, "expected a group key handle"); return NULL; }
    uint32_t nv, np, dom, world; int dist;
    L.group_pkey_info(gk->k, &nv, &np, &dom, &world, &dist);
    napi_create_object(env, &o);
    napi_create_uint32(env, nv, &v); napi_set_named_property(env, o, "nVars", v);
    napi_create_uint32(env, np, &v); napi_set_named_property(env, o, "nPublic", v);
    napi_create_uint32(env, dom, &v); napi_set_named_property(env, o, "domainSize", v);
    napi_create_uint32(env, world, &v); napi_set_named_property(env, o, "world", v);
    napi_get_boolean(env, dist != 0, &v); napi_set_named_property(env, o, "distributedCalcH", v);
    return o;
}

/* ---- bin2int / bin2g1 / bin2g2 of the reference (src/bn128.js:319-351, 714-718) in native code ----
 * proofToObject(ArrayBuffer 384) -> {pi_a: [x, y, z], pi_b: [[x0, x1], [y0, y1], [z0, z1]], pi_c: [x, y, z]} of decimal strings */
static void le256_to_decimal(const uint8_t* le, char* out /* >= 80 bytes */) {
    uint32_t w[8];
    char rev[96];
    int n = 0;
    for (int i = 0; i < 8; i++) w[i] = (uint32_t)le[4 * i] | ((uint32_t)le[4 * i + 1] << 8) | ((uint32_t)le[4 * i + 2] << 16) | ((uint32_t)le[4 * i + 3] << 24);
    for (;;) {
        uint64_t rem = 0;
        int nonzero = 0;
        for (int i = 7; i >= 0; i--) {                 /* w /= 10^9 */
            const uint64_t cur = (rem << 32) | w[i];
            w[i] = (uint32_t)(cur / 1000000000u);
            rem = cur % 1000000000u;
            nonzero |= w[i] != 0;
        }
        for (int k = 0; k < 9; k++) { rev[n++] = (char)('0' + rem % 10); rem /= 10; if (!nonzero && rem == 0) break; }
        if (!nonzero) break;
    }
    while (n > 1 && rev[n - 1] == '0') n--;
    for (int i = 0; i < n; i++) out[i] = rev[n - 1 - i];
    out[n] = 0;
}
static napi_value js_proof_to_object(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    uint8_t* p = NULL; size_t n = 0;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 1 || !get_bytes(env, argv[0], &p, &n) || n != 384) { napi_throw_type_error(env, NULL, "expected the 384 proof bytes"); return NULL; }
    napi_value v[12], o, a, b, c, pair;
    char dec[96];
    for (int i = 0; i < 12; i++) { le256_to_decimal(p + 32 * i, dec); napi_create_string_utf8(env, dec, NAPI_AUTO_LENGTH, &v[i]); }
    napi_create_object(env, &o);
    napi_create_array_with_length(env, 3, &a);
    for (uint32_t i = 0; i < 3; i++) napi_set_element(env, a, i, v[i]);
    napi_create_array_with_length(env, 3, &b);
    for (uint32_t i = 0; i < 3; i++) {
        napi_create_array_with_length(env, 2, &pair);
        napi_set_element(env, pair, 0, v[3 + 2 * i]);
        napi_set_element(env, pair, 1, v[4 + 2 * i]);
        napi_set_element(env, b, i, pair);
    }
    napi_create_array_with_length(env, 3, &c);
    for (uint32_t i = 0; i < 3; i++) napi_set_element(env, c, i, v[9 + i]);
    napi_set_named_property(env, o, "pi_a", a);
    napi_set_named_property(env, o, "pi_b", b);
    napi_set_named_property(env, o, "pi_c", c);
    return o;
}

/* init(device) -> device info string */
static napi_value js_init(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1], s;
    int32_t dev = -1;
    char err[4600];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc > 0) napi_get_value_int32(env, argv[0], &dev);
    if (load_lib(err, sizeof err)) { napi_throw_error(env, NULL, err); return NULL; }
    int rc = L.init(dev);
    if (rc) {
        char msg[600];
        snprintf(msg, sizeof msg, "wsnark_init failed (%d): %s -- there is no CPU fallback", rc, L.last_error());
        napi_throw_error(env, NULL, msg);
        return NULL;
    }
    napi_create_string_utf8(env, L.device_info(), NAPI_AUTO_LENGTH, &s);
    return s;
}
static napi_value js_shutdown(napi_env env, napi_callback_info info) {
    (void)info;
    if (L.h) L.shutdown();
    napi_value u; napi_get_undefined(env, &u);
    return u;
}

static napi_value module_init(napi_env env, napi_value exports) {
    Dl_info di;
    snprintf(L.dir, sizeof L.dir, ".");
    if (dladdr((void*)module_init, &di) && di.dli_fname) {
        snprintf(L.dir, sizeof L.dir, "%s", di.dli_fname);
        char* sl = strrchr(L.dir, '/');
        if (sl) *sl = 0;
    }
    napi_property_descriptor props[] = {
        {"init", NULL, js_init, NULL, NULL, NULL, napi_default, NULL},
        {"shutdown", NULL, js_shutdown, NULL, NULL, NULL, napi_default, NULL},
        {"g1Multiexp", NULL, js_g1, NULL, NULL, NULL, napi_default, NULL},
        {"g2Multiexp", NULL, js_g2, NULL, NULL, NULL, napi_default, NULL},
        {"fft", NULL, js_fft, NULL, NULL, NULL, napi_default, NULL},
        {"calcH", NULL, js_calch, NULL, NULL, NULL, napi_default, NULL},
        {"loadKey", NULL, js_loadkey, NULL, NULL, NULL, napi_default, NULL},
        {"loadKeyFile", NULL, js_loadkey_file, NULL, NULL, NULL, napi_default, NULL},
        {"keyFileInfo", NULL, js_keyfile_info, NULL, NULL, NULL, napi_default, NULL},
        {"groupLoadKeyFile", NULL, js_group_loadkey_file, NULL, NULL, NULL, napi_default, NULL},
        {"hashBytes", NULL, js_hash, NULL, NULL, NULL, napi_default, NULL},
        {"keyInfo", NULL, js_keyinfo, NULL, NULL, NULL, napi_default, NULL},
        {"allocPinned", NULL, js_alloc_pinned, NULL, NULL, NULL, napi_default, NULL},
        {"prove", NULL, js_prove, NULL, NULL, NULL, napi_default, NULL},
        {"waitTables", NULL, js_wait_tables, NULL, NULL, NULL, napi_default, NULL},
        {"verify", NULL, js_verify, NULL, NULL, NULL, napi_default, NULL},
        {"proofToObject", NULL, js_proof_to_object, NULL, NULL, NULL, napi_default, NULL},
        {"loadPoints", NULL, js_points_load, NULL, NULL, NULL, napi_default, NULL},
        {"pointsMultiexp", NULL, js_points_msm, NULL, NULL, NULL, napi_default, NULL},
        {"groupCreate", NULL, js_group_create, NULL, NULL, NULL, napi_default, NULL},
        {"groupFree", NULL, js_group_free, NULL, NULL, NULL, napi_default, NULL},
        {"groupMultiexp", NULL, js_group_msm, NULL, NULL, NULL, napi_default, NULL},
        {"groupLoadKey", NULL, js_group_loadkey, NULL, NULL, NULL, napi_default, NULL},
        {"groupProve", NULL, js_group_prove, NULL, NULL, NULL, napi_default, NULL},
        {"groupWaitTables", NULL, js_group_wait_tables, NULL, NULL, NULL, napi_default, NULL},
        {"groupKeyInfo", NULL, js_group_keyinfo, NULL, NULL, NULL, napi_default, NULL},
    };
    CHECK(env, napi_define_properties(env, exports, sizeof props / sizeof props[0], props));
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, module_init)

// keyfile.hip -- proving keys read from FILES, and the container for keys beyond 4 GiB.
//
// The reference's proving_key.bin (/root/reference tools/buildpkey.js:124-186) addresses its sections with u32 byte offsets
// (:133-139): a key of more than 4 GiB -- about 2^23 constraints -- cannot be written, and BASELINE config 5
// is 2^24.  WSNARK64 is the same file with 64-bit offsets, sections in the same order, byte for byte the
// same contents:
//
//      0  char[8]  "WSNARK64"
//      8  u32      version = 1
//     12  u32      header bytes = 608 (where the first section may start)
//     16  u32      nVars, nPublic, domainSize, 0
//     32  u64      pPolsA, lenPolsA, pPolsB, lenPolsB, pPointsA, pPointsB1, pPointsB2, pPointsC, pHExps     (byte offsets / lengths)
//    104  u64      length of the whole file
//    112  ...      zero
//    160  alfa1, beta1, delta1 (3 x 64 B), beta2, delta2 (2 x 128 B)          (tools/buildpkey.js:141-164)
//    608  polsA, polsB, A[nVars] x 64, B1[nVars] x 64, B2[nVars] x 128, C[nVars-nPublic-1] x 64, hExps[domain] x 64
//         (:166-186), each section at a multiple of 4096
//
// wasmsnark_amd/formats.py and js/formats.js write it (from proving_key.bin, from snarkjs JSON sections, or section by section).
// The loader maps the file read-only and hands pkey_load_sections pointers INTO the map: a rank of N reads only its 1 / N of the five
// point sections (plus the two matrices), and every range is handed back to the kernel (madvise) as soon as it has been copied to
// the staging ring, so the load's resident set is a few tens of MiB whatever the key's size -- not 8 processes x 7.8 GB.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "internal.h"

namespace wsnark {

KeyFile::~KeyFile() {
    if (base) munmap(const_cast<uint8_t*>(base), len);
    if (fd >= 0) close(fd);
}

// hand the whole pages inside [p, p + n) back (the partial pages at either end may still be needed by a neighbouring range)
static void release_pages(const void* p, size_t n) {
    const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t lo = ((uintptr_t)p + pg - 1) & ~(pg - 1), hi = ((uintptr_t)p + n) & ~(pg - 1);
    if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_DONTNEED);
}

static bool range_ok(uint64_t off, uint64_t bytes, uint64_t len) { return off <= len && bytes <= len - off; }

static int parse_container(const uint8_t* b, size_t len, KeySections* S) {
    if (len < 608) { set_last_error("key container shorter than its header"); return WS_ERR_FORMAT; }
    uint32_t w[6];
    memcpy(w, b + 8, 24);
    if (w[0] != 1 || w[1] < 608) { set_last_error("key container: unknown version or header size"); return WS_ERR_FORMAT; }
    const uint32_t nv = w[2], np = w[3], dom = w[4];
    uint64_t q[10];
    memcpy(q, b + 32, 80);
    const uint64_t pA_ = q[0], lA = q[1], pB_ = q[2], lB = q[3], pA = q[4], pB1 = q[5], pB2 = q[6], pC = q[7], pH = q[8], flen = q[9];
    if (flen != len) { set_last_error("key container: the file is not as long as its header says (truncated?)"); return WS_ERR_FORMAT; }
    if (nv == 0 || (uint64_t)np + 1 > nv) { set_last_error("proving key: nPublic + 1 > nVars"); return WS_ERR_FORMAT; }
    const uint64_t nC = (uint64_t)nv - np - 1;
    if (pA_ < w[1] || !range_ok(pA_, lA, len) || !range_ok(pB_, lB, len) || !range_ok(pA, (uint64_t)nv * 64, len) ||
        !range_ok(pB1, (uint64_t)nv * 64, len) || !range_ok(pB2, (uint64_t)nv * 128, len) || !range_ok(pC, nC * 64, len) ||
        !range_ok(pH, (uint64_t)dom * 64, len)) {
        set_last_error("key container: section offsets out of range");
        return WS_ERR_FORMAT;
    }
    *S = KeySections{nv, np, dom, b + 160, b + 224, b + 288, b + 352, b + 480,
                     b + pA_, lA, b + pB_, lB, b + pA, b + pB1, b + pB2, b + pC, b + pH,
                     (uint64_t)nv * 64, (uint64_t)nv * 64, (uint64_t)nv * 128, nC * 64, (uint64_t)dom * 64};
    return WS_OK;
}

int keyfile_open(const char* path, KeyFile* F, KeySections* S) {
    if (!path || !F || !S) return WS_ERR_ARG;
    F->fd = open(path, O_RDONLY | O_CLOEXEC);
    if (F->fd < 0) { set_last_error(std::string("key file: cannot open ") + path + ": " + strerror(errno)); return WS_ERR_ARG; }
    struct stat st;
    if (fstat(F->fd, &st) != 0 || st.st_size < 8) { set_last_error(std::string("key file: cannot stat / too short: ") + path); return WS_ERR_FORMAT; }
    F->len = (size_t)st.st_size;
    void* m = mmap(nullptr, F->len, PROT_READ, MAP_PRIVATE, F->fd, 0);
    if (m == MAP_FAILED) { set_last_error(std::string("key file: mmap failed: ") + strerror(errno)); return WS_ERR_HIP; }
    F->base = (const uint8_t*)m;
    int rc;
    if (memcmp(F->base, "WSNARK64", 8) == 0) {
        F->format = 2;
        rc = parse_container(F->base, F->len, S);
    } else {
        F->format = 1;                                   // the reference's own file (its first word is nVars: never "WSNA")
        rc = pkey_parse(F->base, F->len, S);
    }
    if (rc == WS_OK) S->release = release_pages;
    return rc;
}

}  // namespace wsnark

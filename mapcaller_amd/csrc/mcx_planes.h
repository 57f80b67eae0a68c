// mapcaller_amd/csrc/mcx_planes.h — the layout of the alignment profile's counter planes in HBM.
//
// The reference keeps one 16-byte bit-field record per genome position (MappingRecord_t, src/structure.h:152-163: A C G T
// multi_hit 12 bits each, readCount 4, F1 R2 F2 R1 16 bits each).  Here every counter is a plane of its own, so that GPUs can
// sum them with plain reduces — in the width the counter needs, 22 bytes per position (round 4: ten u32 planes, 40 bytes;
// 124 GB at 3.1 Gbp, which left no room for the pair records and full batches beside them):
//
//   multi_hit                        u32 [stride]   (a repeat's positions collect one count per multi-mapped read and candidate: no
//                                                    bound below 2^16 can be given, and the field saturates — it does not wrap)
//   A C G T readCount F1 R2 F2 R1    u16 [stride] each, in this order
//
// stride = GenomeSize rounded up to 64 positions (every plane starts on a 128-byte line).  16 bits are exact for the nine:
//   * A C G T count the admitted reads over a position with that base there: at most iMaxDuplicate (<= 15, main.cpp:323) reads
//     are admitted per start position and a read is at most 1000 bases long, so a count stays below 15 000 — over all shards of a
//     run too, because the duplicate cap is decided across them.  The 12-bit saturation is applied at the end (k_prof_finalize).
//   * F1 R2 F2 R1 are 16-bit fields that wrap in the reference: arithmetic modulo 2^16 IS their definition.
//   * readCount stops at iMaxDuplicate.
// Two neighbouring positions of a plane share a 32-bit word, and the atomics are word atomics: +1 / -1 on a half is +- 1 or
// +- 65536 on the word.  A plane that holds counts never borrows.  A plane kept as DIFFERENCES while a run is mapped (the strand
// planes, the exact-seed coverage: mcx_profile.h) can hold -1 in its low half, which has then borrowed from the high one: the
// word is L + 65536 H modulo 2^32 with the true signed halves L and H, and because |L| < 2^15 (same bound as above) both come
// back exactly — L = (int16) word, H = (word - L) >> 16 (planes_decode) — before the differences are scanned into counts.
#ifndef MCX_PLANES_H
#define MCX_PLANES_H
#include "mcx_types.h"

namespace mcx {

enum { kPlA = 0, kPlC, kPlG, kPlT, kPlMulti, kPlReadCount, kPlF1, kPlR2, kPlF2, kPlR1, kPlanes }; // plane numbers: the order of the reference's fields
constexpr int kHalfPlanes = 9;

static inline MCX_HD uint64_t planes_stride(int64_t G) { return ((uint64_t)G + 63) & ~(uint64_t)63; }
static inline MCX_HD uint64_t planes_bytes(int64_t G) { return planes_stride(G) * (4 + 2 * kHalfPlanes); }
static inline MCX_HD int planes_slot(int k) { return k < kPlMulti ? k : k - 1; } // which of the nine u16 planes plane k is (k != kPlMulti)

struct PlanesView {
    uint32_t *multi;   // [stride]
    uint16_t *half;    // [9][stride]
    uint64_t stride;
    int64_t G;
    MCX_HD uint16_t *h(int k) const { return half + (uint64_t)planes_slot(k) * stride; }
    MCX_HD uint32_t get(int k, int64_t g) const { return k == kPlMulti ? multi[g] : (uint32_t)h(k)[g]; }
};

static inline MCX_HD PlanesView planes_view(void *base, int64_t G)
{
    PlanesView v;
    v.stride = planes_stride(G); v.G = G;
    v.multi = (uint32_t *)base;
    v.half = (uint16_t *)((uint8_t *)base + v.stride * 4);
    return v;
}

// the two true differences of a word of a difference plane (see above), each modulo 2^16, back in their halves
static inline MCX_HD uint32_t planes_decode(uint32_t w)
{
    const int32_t lo = (int32_t)(int16_t)(w & 0xFFFFu);
    const uint32_t hi = (w - (uint32_t)lo) >> 16;
    return ((uint32_t)lo & 0xFFFFu) | (hi << 16);
}

#if defined(__HIPCC__)
// +1 / -1 on element i of a u16 plane (a word atomic on the word that holds it)
static __device__ __forceinline__ void half_inc(uint16_t *plane, uint64_t i) { atomicAdd((uint32_t *)plane + (i >> 1), (i & 1) ? 0x10000u : 1u); }
static __device__ __forceinline__ void half_dec(uint16_t *plane, uint64_t i) { atomicSub((uint32_t *)plane + (i >> 1), (i & 1) ? 0x10000u : 1u); }
#endif

} // namespace mcx
#endif

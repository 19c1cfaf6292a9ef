// dmr_sort.hpp -- the per-tile sort by (depth_bits, face_id) as a device function: k_sort_tiles (dmr_binning.hip, the
// tet path) runs it as a kernel of its own, k_tri_forward (dmr_tri.hip) runs it at the start of its tile's workgroup.
//    n <= SORT_LDS_KEYS: order-preserving buckets on the depth bits + a rank inside the bucket (round 3), or -- depths bunched in
//    a few buckets, n <= 64 -- segmented rank sort + merge by binary search, in LDS; longer lists: a normalised bitonic network
//    (every compare-exchange ascending: with the list virtually padded by +inf to a power of two, exchanges whose upper
//    index is >= n are no-ops, so any n works without padding storage) in place in global memory.
// Replaces cub::DeviceRadixSort::SortPairs over all (tile | depth) keys (rasterizer_impl.cu:316-324): within a (view, tile)
// a face occurs once, so (depth_bits, face_id) is a total order equal to "stable by key, ties in emission order" (Q6).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmr {

constexpr int SORT_LDS_KEYS = 2048;  // 16 KiB of keys + 8 KiB of ranks
#ifndef DMR_SORT_SEG
#define DMR_SORT_SEG 128
#endif
constexpr int SORT_SEG = DMR_SORT_SEG;  // keys per rank-sorted segment (two per lane)
constexpr int SORT_LDS_BYTES = SORT_LDS_KEYS * (int)(sizeof(uint64_t) + sizeof(uint32_t));

template <class Ptr>
__device__ __forceinline__ void bitonic_pass(Ptr a, uint32_t n, uint32_t npow2, uint32_t tid, uint32_t nthreads, bool global_mem) {
    for (uint32_t k = 2; k <= npow2; k <<= 1) {
        // flip step: partner = i ^ (k - 1)
        for (uint32_t t = tid; t < npow2 / 2; t += nthreads) {
            uint32_t i = ((t & ~(k / 2 - 1)) << 1) | (t & (k / 2 - 1));
            uint32_t l = i ^ (k - 1);
            if (l < n) {
                uint64_t x = a[i], y = a[l];
                if (x > y) { a[i] = y; a[l] = x; }
            }
        }
        if (global_mem) __threadfence_block();
        __syncthreads();
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < npow2 / 2; t += nthreads) {
                uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                uint32_t l = i | j;
                if (l < n) {
                    uint64_t x = a[i], y = a[l];
                    if (x > y) { a[i] = y; a[l] = x; }
                }
            }
            if (global_mem) __threadfence_block();
            __syncthreads();
        }
    }
}

// One 256-thread workgroup sorts the n keys at keys[begin, begin + n) and writes their low words (the face ids) to
// face_list[begin, begin + n).  s_keys / s_rank: SORT_LDS_KEYS entries of LDS each, free on entry (the caller has passed a
// barrier since their last use) and free again after the caller's next barrier.  n > 0.
__device__ __forceinline__ void sort_tile(uint32_t begin, uint32_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ face_list,
                                          uint64_t* __restrict__ s_keys, uint32_t* __restrict__ s_rank, uint32_t tid) {
    const uint32_t wave = tid >> 6, lane = tid & 63;
#ifndef DMR_SORT_BUCKETS
#define DMR_SORT_BUCKETS 1
#endif
    // BUCKETS FIRST (round 3).  The rank sort below compares every key with the 128 keys of its segment and then searches every
    // other segment: ~400 VALU instructions per key, a fifth of k_tri_forward's instructions in a kernel whose VALUs are busy 97 %
    // of the time.  Keys that are cut into SORT_NB order-preserving buckets first -- by the bits of
    // (depth - the tile's smallest depth) just below the range's leading bit -- only have to be ranked inside their bucket: a key
    // claims a slot of its bucket with a returning LDS atomic, one wave scans the bucket sizes, the keys are rewritten bucket by
    // bucket, and a key's place is its bucket's base + the number of smaller keys IN the bucket (a loop as long as the wave's
    // fullest bucket instead of 128 + two 7-step searches at C4).  No merge; the same total order, so the same face_list.
    // A tile whose fullest bucket holds more than SORT_BUCKET_MAX keys (depths bunched in a few buckets) is sorted as before.
    // k_tri_forward 77.0 -> 73.2 us at C4 on the same box; 64 / 128 / 256 buckets and a limit of 64 / 96 / 160 within 1 us of
    // each other (profiles/r03/variants_sort_buckets_c4.txt).
#ifndef DMR_SORT_NB_LOG2
#define DMR_SORT_NB_LOG2 6
#endif
#ifndef DMR_SORT_BUCKET_MAX
#define DMR_SORT_BUCKET_MAX 96
#endif
    constexpr uint32_t SORT_NB_LOG2 = DMR_SORT_NB_LOG2, SORT_NB = 1u << SORT_NB_LOG2, SORT_BUCKET_MAX = DMR_SORT_BUCKET_MAX;
    static_assert(SORT_NB >= 64 && SORT_NB <= 512, "a wave scans the bucket sizes, SORT_NB / 64 per lane");
    if (DMR_SORT_BUCKETS && n <= SORT_LDS_KEYS && n > 64u) {
        uint32_t* const s_cnt = s_rank;            // [SORT_NB] bucket sizes, then claim cursors are not needed again
        uint32_t* const s_base = s_rank + SORT_NB; // [SORT_NB] exclusive scan
        uint32_t* const s_misc = s_rank + 2 * SORT_NB;  // [0] min depth bits, [1] max depth bits, [2] fullest bucket
        constexpr int SLOTS = SORT_LDS_KEYS / 256;
        uint64_t mk[SLOTS];
        uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int q = 0; q < SLOTS; q++) {
            const uint32_t i = tid + 256u * q;
            mk[q] = i < n ? keys[begin + i] : ~0ull;
            if (i < n) { const uint32_t d = (uint32_t)(mk[q] >> 32); lo = d < lo ? d : lo; hi = d > hi ? d : hi; }
        }
        for (uint32_t i = tid; i < SORT_NB; i += 256u) s_cnt[i] = 0u;
        if (tid == 0) { s_misc[0] = 0xffffffffu; s_misc[1] = 0u; }
        __syncthreads();
#pragma unroll
        for (int dlt = 32; dlt > 0; dlt >>= 1) {
            const uint32_t ol = (uint32_t)__shfl_xor((int)lo, dlt, 64), oh = (uint32_t)__shfl_xor((int)hi, dlt, 64);
            lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi;
        }
        if (lane == 0) { atomicMin(&s_misc[0], lo); atomicMax(&s_misc[1], hi); }
        __syncthreads();
        const uint32_t mn = s_misc[0], range = s_misc[1] - mn;
        // shift: the range's leading bit lands on bit log2(SORT_NB) - 1, so (d - mn) >> shift < SORT_NB
        const uint32_t bits = range ? 32u - (uint32_t)__clz((int)range) : 0u;
        const uint32_t shift = bits > SORT_NB_LOG2 ? bits - SORT_NB_LOG2 : 0u;
        uint32_t where[SLOTS];  // bucket << 16 | slot inside the bucket
#pragma unroll
        for (int q = 0; q < SLOTS; q++) {
            const uint32_t i = tid + 256u * q;
            where[q] = 0u;
            if (i < n) {
                const uint32_t b = ((uint32_t)(mk[q] >> 32) - mn) >> shift;
                where[q] = (b << 16) | atomicAdd(&s_cnt[b], 1u);
            }
        }
        __syncthreads();
        if (tid < 64u) {  // one wave: exclusive scan of the bucket sizes (SORT_NB / 64 per lane), and the fullest bucket
            constexpr uint32_t PER = SORT_NB / 64u;
            uint32_t c[PER], sum = 0u, mx = 0u;
#pragma unroll
            for (uint32_t i = 0; i < PER; i++) { c[i] = s_cnt[tid * PER + i]; sum += c[i]; mx = c[i] > mx ? c[i] : mx; }
            uint32_t incl = sum;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, dlt, 64);
                if (lane >= (uint32_t)dlt) incl += o;
            }
#pragma unroll
            for (int dlt = 32; dlt > 0; dlt >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, dlt, 64); mx = o > mx ? o : mx; }
            uint32_t run = incl - sum;
#pragma unroll
            for (uint32_t i = 0; i < PER; i++) { s_base[tid * PER + i] = run; run += c[i]; }
            if (tid == 0) s_misc[2] = mx;
        }
        __syncthreads();
        if (s_misc[2] <= SORT_BUCKET_MAX) {  // uniform
#pragma unroll
            for (int q = 0; q < SLOTS; q++) {
                const uint32_t i = tid + 256u * q;
                if (i < n) s_keys[s_base[where[q] >> 16] + (where[q] & 0xffffu)] = mk[q];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < SLOTS; q++) {
                const uint32_t i = tid + 256u * q;
                if (i >= n) continue;
                const uint32_t b = where[q] >> 16, b0 = s_base[b], b1 = b0 + s_cnt[b];
                const uint64_t key = mk[q];
                uint32_t r = b0;
                for (uint32_t j = b0; j < b1; j++) r += s_keys[j] < key ? 1u : 0u;
                face_list[begin + r] = (uint32_t)key;
            }
            return;
        }
        __syncthreads();  // (the LDS is read again from the start below)
    }
    if (n <= SORT_LDS_KEYS) {
        // Segmented rank sort + merge.  Keys are unique, so a key's output slot is the number of smaller keys.  The n keys
        // are cut into S segments of SORT_SEG = 128; wave w rank-sorts segments w, w + 4, ... on its own (lane l holds keys
        // l and l + 64 of the segment and streams all of it as LDS broadcasts: len reads, 2 len compares per lane); the
        // segments are then rewritten in sorted order, and every key adds to its rank in its own segment the number of
        // smaller keys in each other segment -- a 7-step binary search.  Work: n * 128 compares + (S - 1) * 7 reads per key,
        // against n^2 compares of one rank sort over the tile (C4's tiles average 315 keys, the longest 642; history: every
        // thread looping over all keys 48 us; one rank sort per tile, split over the four waves, 24 us; tiles above 512 keys
        // took a bitonic network in LDS, ~55 barrier-separated stages).
        const uint32_t S = (n + SORT_SEG - 1u) / SORT_SEG;
        for (uint32_t i = tid; i < n; i += 256) s_keys[i] = keys[begin + i];
        __syncthreads();
        for (uint32_t sg = wave; sg < S; sg += 4u) {
            const uint32_t b0 = sg * SORT_SEG, len = min((uint32_t)SORT_SEG, n - b0);
            const uint64_t k0 = lane < len ? s_keys[b0 + lane] : ~0ull, k1 = lane + 64u < len ? s_keys[b0 + lane + 64u] : ~0ull;
            uint32_t r0 = 0u, r1 = 0u;
#pragma unroll 8
            for (uint32_t j = 0; j < len; j++) {
                const uint64_t kj = s_keys[b0 + j];
                r0 += kj < k0 ? 1u : 0u; r1 += kj < k1 ? 1u : 0u;
            }
            if (lane < len) s_rank[b0 + lane] = r0;
            if (lane + 64u < len) s_rank[b0 + lane + 64u] = r1;
        }
        __syncthreads();
        // every segment sorted in place (through registers: a thread holds at most 8 keys)
        uint64_t mk[SORT_LDS_KEYS / 256]; uint32_t mr[SORT_LDS_KEYS / 256];
#pragma unroll
        for (int q = 0; q < SORT_LDS_KEYS / 256; q++) {
            const uint32_t i = tid + 256u * q;
            mk[q] = i < n ? s_keys[i] : 0ull; mr[q] = i < n ? s_rank[i] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SORT_LDS_KEYS / 256; q++) {
            const uint32_t i = tid + 256u * q;
            if (i < n) s_keys[(i & ~(uint32_t)(SORT_SEG - 1)) + mr[q]] = mk[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SORT_LDS_KEYS / 256; q++) {
            const uint32_t i = tid + 256u * q;
            if (i >= n) continue;
            const uint64_t key = s_keys[i];
            const uint32_t own = i / SORT_SEG;
            uint32_t rank = i - own * SORT_SEG;
            for (uint32_t sg = 0; sg < S; sg++) {
                if (sg == own) continue;
                const uint32_t b0 = sg * SORT_SEG;
                uint32_t lo = 0, hi = min((uint32_t)SORT_SEG, n - b0);  // lower bound of `key` in the sorted segment
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_keys[b0 + mid] < key) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
            face_list[begin + rank] = (uint32_t)key;  // nothing reads the sorted keys
        }
    } else {
        uint64_t* g = keys + begin;
        uint32_t npow2 = 1;
        while (npow2 < n) npow2 <<= 1;
        bitonic_pass(g, n, npow2, tid, 256, true);
        for (uint32_t i = tid; i < n; i += 256) face_list[begin + i] = (uint32_t)g[i];
    }
}

}  // namespace dmr

// dmr_sort.hpp -- the per-tile sort by (depth_bits, face_id) as a device function: k_sort_tiles (dmr_binning.hip, the
// tet path) runs it as a kernel of its own, k_tri_forward (dmr_tri.hip) runs it at the start of its tile's workgroup.
//    n <= SORT_LDS_KEYS: segmented rank sort + merge by binary search, in LDS; longer lists: a normalised bitonic network
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

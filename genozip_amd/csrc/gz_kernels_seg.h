// gz_kernels_seg.h -- the seg-side appends of a context, a whole column at a time
//
// The reference evaluates the snips of a context one by one (ctx_create_node_do src/context.c:320-404): look the snip
// up in the dictionary cloned from the file (ol_nodes), then in the VBlock's own hash table (hash_get_entry_for_seg
// src/hash.c:530-576, hash_do src/hash.h:30-52), append it to dict / nodes if it is new (ctx_insert_to_dict
// src/context.c:50-71), count it, and append the node index to the b250 (b250_seg_append src/b250.c:112-163). What that
// leaves behind depends only on the ORDER OF FIRST OCCURRENCES:
//     node index of a snip = its index in ol_nodes, else ol_nodes.len + (rank of its first occurrence in the VBlock)
// so the column is done data-parallel:
//   k_col_clear      counts = 0, result block
//   k_col_insert_ol  the cloned dictionary into an open-addressing table (ids 0 .. n_ol-1)
//   k_col_insert     every snip (id n_ol + position): claim the slot of its string or join it; the slot keeps the
//                    SMALLEST id among equal strings (atomicMin) = the ol node if there is one, else the first occurrence
//   k_col_first      representative of every snip; first occurrences flagged; per-tile (count, dict bytes)
//   k_col_scan_a     per column: exclusive scan of the tile sums -> n_new, dict_len
//   k_col_assign     rank and dict offset of every first occurrence (workgroup prefix sum); nodes; dict bytes + NUL
//   k_col_node       node index of every snip; all-the-same; per-tile b250 bytes
//   k_col_counts     occurrences per node (LDS table per 16 K snips, then global atomics)
//   k_col_scan_b     per column: scan -> b250_len (or the single entry of an all-the-same column)
//   k_col_b250       seg-format words: little endian, type tag in the LAST byte, nodes new to the VBlock always 4 bytes
// The table is this library's own (never written to the file); the snip mixing function is the reference's rotate-xor,
// folded to a power-of-two table by a multiplication instead of its modulo-a-prime.
//
// dyn_int_append (src/dyn_int.c:232-320) likewise: the final width is a function of the column's min / max
// (k_dyn_minmax, k_dyn_decide), the values are then narrowed in one pass (k_dyn_write). And seg_add_to_local_fixed_do
// (src/seg.c:1268-1287) over a column is a gather of the snips into one blob (k_blob_sum/scan/copy) - the "transpose"
// of a field of the line buffer into its context's local.
#pragma once
#include "gz_device.h"
#include "gz_devutil.h"
#include <gz_intrin.h>
#include "gz_kernels_ctx.h"

#define GZ_COL_TILE 256

struct GzdColumn {
    const uint8_t *text; const uint32_t *off, *len; uint32_t n;
    const uint8_t *ol_dict; const uint64_t *ol_char_index; const uint32_t *ol_snip_len; uint32_t n_ol;
    int32_t *node_index; uint8_t *dict; uint64_t dict_cap; uint64_t *node_char_index; uint32_t *node_snip_len;
    uint32_t *counts; uint8_t *b250; GzColumnResult *result;
    // scratch
    uint32_t *table; uint32_t table_bits;
    uint32_t *rep;            // [n] slot of the snip, then id of its representative (0xffffffff: no snip)
    uint32_t *rank;           // [n] rank of the first occurrence at this position among the VBlock's new nodes
    uint64_t *tile_a;         // [tiles] (dict bytes << 32 | first occurrences) per tile, then their exclusive scan (a VBlock's dictionary stays below 4 GB)
    uint64_t *tile_b;         // [tiles] b250 bytes per tile, then their exclusive scan
    uint32_t *not_same;       // set when two entries differ
};

// ---- workgroup helpers (256 threads; use the first 2 KB + 16 bytes of gz_lds) -------------------------------------
// exclusive prefix sum of v over the threads of the workgroup; *total = sum over all threads
__device__ static inline uint64_t d_wg_scan_u64 (uint64_t v, int tid, uint64_t *total)
{
    uint64_t *sh = (uint64_t *)gz_lds;
    __syncthreads ();                                   // (the LDS may still be in use by the caller's previous scan)
    sh[tid] = v;
    __syncthreads ();
    for (int d = 1; d < 256; d <<= 1) {
        const uint64_t add = tid >= d ? sh[tid - d] : 0;
        __syncthreads ();
        sh[tid] += add;
        __syncthreads ();
    }
    const uint64_t incl = sh[tid];
    *total = sh[255];
    return incl - v;
}

// in-place exclusive scan of tiles[0 .. n_tiles) by one workgroup; returns the total. Every thread takes 16 consecutive
// entries (one 128-byte line) per round, so a round covers 4096 entries with ONE workgroup scan (the newline tiles of
// a 1.6 GB text are 98 000 entries: 24 rounds instead of 383).
#define GZ_SCAN_PER 16
__device__ static inline uint64_t d_wg_scan_array (uint64_t *tiles, uint32_t n_tiles, int tid)
{
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n_tiles; base += 256 * GZ_SCAN_PER) {
        const uint32_t i0 = base + (uint32_t)tid * GZ_SCAN_PER;
        uint64_t mine = 0;
        for (uint32_t k = 0; k < GZ_SCAN_PER && i0 + k < n_tiles; k++) mine += tiles[i0 + k];
        uint64_t total;
        uint64_t run = carry + d_wg_scan_u64 (mine, tid, &total);
        for (uint32_t k = 0; k < GZ_SCAN_PER && i0 + k < n_tiles; k++) { const uint64_t v = tiles[i0 + k]; tiles[i0 + k] = run; run += v; }
        carry += total;
    }
    return carry;
}

// all lanes of the wave copy len bytes (the arguments are wave-uniform): 4 bytes per lane and step, whatever the
// alignment (256 bytes per step: a 150-byte read or quality string is one step), the last 1..3 bytes one by one
typedef uint32_t gz_u32_unaligned __attribute__((aligned (1)));
// the same bytes through GLOBAL pointers (a load / store through a generic one is a flat instruction: it also counts as an LDS operation)
typedef const __attribute__((address_space(1))) uint8_t *GzGlobalCU8P;
typedef __attribute__((address_space(1))) uint8_t *GzGlobalU8P;
typedef const __attribute__((address_space(1))) gz_u32_unaligned *GzGlobalCU32UP;
typedef __attribute__((address_space(1))) gz_u32_unaligned *GzGlobalU32UP;
__device__ static inline void d_wave_copy (uint8_t *dst, const uint8_t *src, uint32_t len, int lane)
{
    const uint32_t whole = len & ~3u;
    for (uint32_t b = (uint32_t)lane * 4; b < whole; b += 256) *(gz_u32_unaligned *)(dst + b) = *(const gz_u32_unaligned *)(src + b);
    if ((uint32_t)lane < len - whole) dst[whole + lane] = src[whole + lane];
}

// ---- rows a1 + a2 ---------------------------------------------------------------------------------------------------
// hash.h:36-46: rotate-xor of the snip's bytes through a 64-bit word
__device__ static inline uint64_t d_snip_hash (const uint8_t *s, uint32_t len)
{
    uint64_t r = 0;
    for (uint32_t i = 0; i < len; i++) r = ((r << 23) | (r >> 41)) ^ (uint64_t)s[i];
    return r;
}

__device__ static inline const uint8_t *d_col_snip (const GzdColumn &C, uint32_t id, uint32_t *len)
{
    if (id < C.n_ol) { *len = C.ol_snip_len[id]; return C.ol_dict + C.ol_char_index[id]; }
    const uint32_t k = id - C.n_ol;
    *len = C.len[k];
    return C.text + C.off[k];
}

__device__ static inline bool d_same_bytes (const uint8_t *a, const uint8_t *b, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) if (a[i] != b[i]) return false;
    return true;
}

// Equal strings walk the same probe sequence and slots are never vacated, so they all end in ONE slot; every id a slot
// ever holds names the same string, which makes the comparison below independent of the races on the id.
__device__ static inline uint32_t d_col_insert (const GzdColumn &C, uint32_t id, const uint8_t *s, uint32_t len, uint64_t hash)
{
    const uint32_t mask = (1u << C.table_bits) - 1;
    uint32_t slot = (uint32_t)((hash * 0x9E3779B97F4A7C15ull) >> (64 - C.table_bits));
    for (;;) {
        // a plain look first: whatever id it finds - however stale - names the slot's string for good, and a column
        // that is one word ten thousand times over would otherwise be ten thousand atomics on one address
        uint32_t cur = C.table[slot];
        if (cur == 0xffffffffu) {
            cur = atomicCAS (&C.table[slot], 0xffffffffu, id);
            if (cur == 0xffffffffu) return slot;                               // mine
        }
        uint32_t ol;
        const uint8_t *o = d_col_snip (C, cur, &ol);
        if (ol == len && d_same_bytes (o, s, len)) { if (id < cur) atomicMin (&C.table[slot], id); return slot; }
        slot = (slot + 1) & mask;
    }
}

// grid (tiles over n_ol + n, columns)
__global__ void __launch_bounds__(256) k_col_clear (GzdColumn *cols)
{
    const GzdColumn C = cols[blockIdx.y];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (uint64_t)C.n_ol + C.n) C.counts[i] = 0;
    if (!i) {
        GzColumnResult r; r.dict_len = 0; r.b250_len = 0; r.b250_count = C.n; r.n_new = 0; r.all_the_same = 0; r.status = GZ_ST_OK;
        *C.result = r;
        *C.not_same = 0;
    }
}

// grid (tiles over n_ol, columns)
__global__ void __launch_bounds__(256) k_col_insert_ol (GzdColumn *cols)
{
    const GzdColumn C = cols[blockIdx.y];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C.n_ol) return;
    uint32_t len;
    const uint8_t *s = d_col_snip (C, i, &len);
    (void)d_col_insert (C, i, s, len, d_snip_hash (s, len));
}

// grid (tiles over n, columns), and so are the following
__global__ void __launch_bounds__(256) k_col_insert (GzdColumn *cols)
{
    const GzdColumn C = cols[blockIdx.y];
    if (blockIdx.x * 256 >= C.n) return;
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const uint32_t len = k < C.n ? C.len[k] : 0, off = len ? C.off[k] : 0;
    // Columns of few distinct words (most of them) would put a wave's 64 lanes on the same slot: the first lane of every
    // group of equal snips inserts for the group - like the reference's "same as the previous snip" short cut
    // (context.c:338-342). Four groups per wave, whoever is left goes on its own.
    const int lane = threadIdx.x & 63;
    const uint64_t hv = d_snip_hash (C.text + off, len);
    uint32_t slot = 0xffffffffu;
    uint64_t todo = __ballot (len != 0);
    for (int round = 0; round < 4 && todo; round++) {
        const int src = __ffsll ((unsigned long long)todo) - 1;
        const uint32_t lh_lo = (uint32_t)__shfl ((int)(uint32_t)hv, src), lh_hi = (uint32_t)__shfl ((int)(uint32_t)(hv >> 32), src);
        const uint32_t lo = (uint32_t)__shfl ((int)off, src), ll = (uint32_t)__shfl ((int)len, src);
        const bool same = len && slot == 0xffffffffu && (uint32_t)hv == lh_lo && (uint32_t)(hv >> 32) == lh_hi && len == ll
                          && d_same_bytes (C.text + off, C.text + lo, len);
        uint32_t ls = 0;
        if (lane == src) ls = d_col_insert (C, C.n_ol + k, C.text + off, len, hv);
        ls = (uint32_t)__shfl ((int)ls, src);
        if (same) slot = ls;
        const uint64_t group = __ballot (same);
        todo &= ~group;
        if (__popcll (group) < 4) break;                   // (a column of mostly distinct words: the rounds would be wasted)
    }
    if (len && slot == 0xffffffffu) slot = d_col_insert (C, C.n_ol + k, C.text + off, len, hv);
    if (k < C.n) C.rep[k] = slot;
}

__global__ void __launch_bounds__(256) k_col_first (GzdColumn *cols)
{
    const GzdColumn C = cols[blockIdx.y];
    if (blockIdx.x * 256 >= C.n) return;
    const int tid = threadIdx.x;
    const uint32_t k = blockIdx.x * 256 + tid;
    uint64_t v = 0;
    if (k < C.n) {
        const uint32_t slot = C.rep[k];
        uint32_t rep = 0xffffffffu;
        if (slot != 0xffffffffu) {
            rep = C.table[slot];
            if (rep == C.n_ol + k) v = ((uint64_t)(C.len[k] + 1) << 32) | 1;
        }
        C.rep[k] = rep;
    }
    uint64_t total;
    (void)d_wg_scan_u64 (v, tid, &total);
    if (!tid) C.tile_a[blockIdx.x] = total;
}

// grid (columns)
__global__ void __launch_bounds__(256) k_col_scan_a (GzdColumn *cols)
{
    const GzdColumn C = cols[blockIdx.x];
    const uint32_t n_tiles = (C.n + GZ_COL_TILE - 1) / GZ_COL_TILE;
    const uint64_t total = d_wg_scan_array (C.tile_a, n_tiles, threadIdx.x);
    if (!threadIdx.x) {
        C.result->n_new = (uint32_t)total;
        C.result->dict_len = total >> 32;
        if ((total >> 32) > C.dict_cap) C.result->status = GZ_ST_TOO_SMALL;     // (nodes and b250 are still right; dict is not written)
    }
}

__global__ void __launch_bounds__(256) k_col_assign (GzdColumn *cols)
{
    const GzdColumn C = cols[blockIdx.y];
    if (blockIdx.x * 256 >= C.n) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t k = blockIdx.x * 256 + tid;
    const bool first = k < C.n && C.rep[k] == C.n_ol + k;
    const uint32_t len = first ? C.len[k] : 0;
    uint64_t total;
    const uint64_t at = C.tile_a[blockIdx.x] + d_wg_scan_u64 (first ? ((uint64_t)(len + 1) << 32) | 1 : 0, tid, &total);
    const uint32_t rank = (uint32_t)at;
    const uint64_t dict_at = at >> 32;
    if (first) {
        C.rank[k] = rank;
        C.node_char_index[rank] = dict_at;
        C.node_snip_len[rank] = len;
    }
    if (C.result->status == GZ_ST_TOO_SMALL) return;
    // the new snips into the dictionary, each followed by a NUL (context.c:62-65): the wave takes them one at a time
    const uint32_t off = first ? C.off[k] : 0;
    for (uint64_t m = __ballot (first); m; m &= m - 1) {
        const int src = __ffsll ((unsigned long long)m) - 1;
        const uint32_t o = (uint32_t)__shfl ((int)off, src), l = (uint32_t)__shfl ((int)len, src);
        const uint32_t d_lo = (uint32_t)__shfl ((int)(uint32_t)dict_at, src);    // (below 4 GB)
        uint8_t *dst = C.dict + d_lo;
        d_wave_copy (dst, C.text + o, l, lane);
        if (!lane) dst[l] = 0;
    }
}

// seg-format length of a node index (b250.c:151-163, 82-107)
__device__ static inline uint32_t d_seg_word (int32_t node, uint32_t n_ol, uint32_t *code)
{
    if (node >= 0 && (uint32_t)node >= n_ol) { *code = (7u << 29) | (uint32_t)node; return 4; }
    return (uint32_t)d_varl_code (node, code);
}

__device__ static inline int32_t d_col_node_of (const GzdColumn &C, uint32_t k)
{
    const uint32_t rep = C.rep[k];
    if (rep == 0xffffffffu) return C.off[k] == GZ_SNIP_MISSING ? -4 : -3;         // context.c:331-335
    return (int32_t)(rep < C.n_ol ? rep : C.n_ol + C.rank[rep - C.n_ol]);
}

__global__ void __launch_bounds__(256) k_col_node (GzdColumn *cols)
{
    const GzdColumn C = cols[blockIdx.y];
    if (blockIdx.x * 256 >= C.n) return;
    const int tid = threadIdx.x;
    const uint32_t k = blockIdx.x * 256 + tid;
    const int32_t node0 = d_col_node_of (C, 0);
    const bool on = k < C.n;
    int32_t node = node0;
    uint32_t bytes = 0;
    if (on) {
        node = d_col_node_of (C, k);
        C.node_index[k] = node;
        uint32_t code;
        bytes = d_seg_word (node, C.n_ol, &code);
    }
    if (__ballot (node != node0) && !(tid & 63) && !*C.not_same) atomicMax (C.not_same, 1u);   // (half a million waves on one address otherwise)
    uint64_t total;
    (void)d_wg_scan_u64 (bytes, tid, &total);
    if (!tid) C.tile_b[blockIdx.x] = total;
}

// counts (vctx->counts, context.c:355,383): a workgroup takes GZ_COUNT_TILES tiles and collects them in a small
// direct-mapped table in LDS first - a 30-million-entry column of 2 000 words is otherwise 30 million global atomics
// on 2 000 addresses (6.1 ms; 1.5 ms this way). What does not find its place in the table goes to memory directly.
#define GZ_COUNT_TILES 64
#define GZ_COUNT_SLOTS 2048
// grid (tiles / GZ_COUNT_TILES, columns)
__global__ void __launch_bounds__(256) k_col_counts (GzdColumn *cols)
{
    const GzdColumn C = cols[blockIdx.y];
    const uint32_t k0 = blockIdx.x * (GZ_COUNT_TILES * 256);
    if (k0 >= C.n) return;
    uint32_t *key = (uint32_t *)gz_lds, *cnt = key + GZ_COUNT_SLOTS;
    const int tid = threadIdx.x;
    for (int i = tid; i < GZ_COUNT_SLOTS; i += 256) { key[i] = 0xffffffffu; cnt[i] = 0; }
    __syncthreads ();
    for (int t = 0; t < GZ_COUNT_TILES; t++) {
        const uint32_t k = k0 + (uint32_t)t * 256 + tid;
        if (k >= C.n) break;
        const int32_t node = C.node_index[k];
        if (node < 0) continue;                                                // empty / missing are not counted (context.c:331-335)
        const uint32_t h = ((uint32_t)node * 0x9E3779B1u) >> 21;              // 11 bits
        uint32_t cur = key[h];
        if (cur == 0xffffffffu) { cur = atomicCAS (&key[h], 0xffffffffu, (uint32_t)node); if (cur == 0xffffffffu) cur = (uint32_t)node; }
        if (cur == (uint32_t)node) atomicAdd (&cnt[h], 1u);
        else atomicAdd (&C.counts[node], 1u);
    }
    __syncthreads ();
    for (int i = tid; i < GZ_COUNT_SLOTS; i += 256) if (cnt[i]) atomicAdd (&C.counts[key[i]], cnt[i]);
}

// grid (columns)
__global__ void __launch_bounds__(256) k_col_scan_b (GzdColumn *cols)
{
    const GzdColumn C = cols[blockIdx.x];
    const uint32_t n_tiles = (C.n + GZ_COL_TILE - 1) / GZ_COL_TILE;
    const uint64_t total = d_wg_scan_array (C.tile_b, n_tiles, threadIdx.x);
    if (threadIdx.x || !C.n) return;
    if (*C.not_same) { C.result->b250_len = total; return; }
    // all the same: ONE entry, however many times it was appended (b250.c:117-141)
    uint32_t code;
    const uint32_t bytes = d_seg_word (C.node_index[0], C.n_ol, &code);
    for (uint32_t i = 0; i < bytes; i++) C.b250[i] = (uint8_t)(code >> (8 * i));
    C.result->b250_len = bytes;
    C.result->all_the_same = 1;
}

__global__ void __launch_bounds__(256) k_col_b250 (GzdColumn *cols)
{
    const GzdColumn C = cols[blockIdx.y];
    if (blockIdx.x * 256 >= C.n || !*C.not_same) return;
    const int tid = threadIdx.x;
    const uint32_t k = blockIdx.x * 256 + tid;
    uint32_t code = 0, bytes = 0;
    if (k < C.n) bytes = d_seg_word (C.node_index[k], C.n_ol, &code);
    uint64_t total;
    const uint64_t at = C.tile_b[blockIdx.x] + d_wg_scan_u64 (bytes, tid, &total);
    for (uint32_t i = 0; i < bytes; i++) C.b250[at + i] = (uint8_t)(code >> (8 * i));   // little endian: the tag goes last
}

// ---- row a3: dyn_int_append over a column ---------------------------------------------------------------------------
struct GzdDynInt {
    const int64_t *values; const uint8_t *is_nothing; uint64_t n; uint32_t nothing_char;
    uint8_t *out; GzDynIntResult *result;
    int64_t *tile_min, *tile_max;          // scratch [tiles]
    const uint64_t *n_dev;                 // optional: the actual count (<= n) lives on the device
};
__device__ static inline uint64_t d_dyn_n (const GzdDynInt &D) { return D.n_dev ? (*D.n_dev < D.n ? *D.n_dev : D.n) : D.n; }

#define GZ_DYN_TILE 1024                   // values per workgroup

// grid (tiles, columns)
__global__ void __launch_bounds__(256) k_dyn_minmax (GzdDynInt *cols)
{
    const GzdDynInt D = cols[blockIdx.y];
    const uint64_t base = (uint64_t)blockIdx.x * GZ_DYN_TILE;
    if (base >= D.n) return;                                  // (tiles beyond the planning bound do not exist)
    const uint64_t Dn = d_dyn_n (D);
    const int tid = threadIdx.x;
    int64_t mn = INT64_MAX, mx = INT64_MIN;
    for (int j = 0; j < GZ_DYN_TILE / 256; j++) {
        const uint64_t k = base + (uint64_t)j * 256 + tid;
        if (k < Dn && !(D.is_nothing && D.is_nothing[k])) { const int64_t v = D.values[k]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    }
    int64_t *sh = (int64_t *)gz_lds;
    sh[tid] = mn; sh[256 + tid] = mx;
    __syncthreads ();
    for (int d = 128; d; d >>= 1) {
        if (tid < d) {
            if (sh[tid + d] < sh[tid]) sh[tid] = sh[tid + d];
            if (sh[256 + tid + d] > sh[256 + tid]) sh[256 + tid] = sh[256 + tid + d];
        }
        __syncthreads ();
    }
    if (!tid) { D.tile_min[blockIdx.x] = sh[0]; D.tile_max[blockIdx.x] = sh[256]; }
}

// lt_order (dyn_int.c:17) in GZ_LT_* numbering, with the ranges of local_type.h
__device__ static inline int64_t d_order_min (int o) { return o == 2 ? -128 : o == 4 ? -32768 : o == 6 ? -2147483648LL : o == 7 ? INT64_MIN : 0; }
__device__ static inline int64_t d_order_max (int o)
{
    return o == 1 ? 255 : o == 2 ? 127 : o == 3 ? 65535 : o == 4 ? 32767 : o == 5 ? 4294967295LL : o == 6 ? 2147483647LL : INT64_MAX;
}

// grid (columns). The walk of dyn_init_prepare (dyn_int.c:232-282) only ever moves to the first order that holds all
// the values seen so far, so its end state is the first order that holds them all - except that a column whose FIRST
// append is a nothing_char starts with min = max = 0xff (dyn_int.c:327-328), which takes part in every later resize
// (but cannot cause one).
__global__ void __launch_bounds__(256) k_dyn_decide (GzdDynInt *cols)
{
    const GzdDynInt D = cols[blockIdx.x];
    const int tid = threadIdx.x;
    const uint64_t Dn = d_dyn_n (D);
    const uint32_t n_tiles = (uint32_t)((D.n + GZ_DYN_TILE - 1) / GZ_DYN_TILE);   // (tiles past Dn hold the neutral elements)
    int64_t mn = INT64_MAX, mx = INT64_MIN;
    for (uint32_t i = tid; i < n_tiles; i += 256) { if (D.tile_min[i] < mn) mn = D.tile_min[i]; if (D.tile_max[i] > mx) mx = D.tile_max[i]; }
    int64_t *sh = (int64_t *)gz_lds;
    sh[tid] = mn; sh[256 + tid] = mx;
    __syncthreads ();
    if (tid) return;
    for (int i = 1; i < 256; i++) { if (sh[i] < mn) mn = sh[i]; if (sh[256 + i] > mx) mx = sh[256 + i]; }
    const int nc = D.nothing_char != 0;
    int order = 1;
    if (mn <= mx && (mn < 0 || mx > 255 - nc)) {                                // some value does not fit UINT8
        if (Dn && D.is_nothing && D.is_nothing[0]) { if (mn > 0xff) mn = 0xff; if (mx < 0xff) mx = 0xff; }
        for (order = 2; order < 7; order++) if (mn >= d_order_min (order) && mx <= d_order_max (order) - nc) break;
    }
    const int lt[8] = { 0, GZ_LT_UINT8, GZ_LT_INT8, GZ_LT_UINT16, GZ_LT_INT16, GZ_LT_UINT32, GZ_LT_INT32, GZ_LT_INT64 };
    D.result->ltype = lt[order];
    D.result->width = order <= 2 ? 1 : order <= 4 ? 2 : order <= 6 ? 4 : 8;
    D.result->len = Dn * D.result->width;
    D.result->order = (uint32_t)order;
}

// grid (tiles, columns)
__global__ void __launch_bounds__(256) k_dyn_write (GzdDynInt *cols)
{
    const GzdDynInt D = cols[blockIdx.y];
    const uint64_t base = (uint64_t)blockIdx.x * GZ_DYN_TILE;
    const uint64_t Dn = d_dyn_n (D);
    if (base >= Dn) return;
    const uint32_t w = D.result->width;
    const int64_t top = d_order_max ((int)D.result->order);                    // a nothing_char is the type's maximum (dyn_int.c:334-341)
    for (int j = 0; j < GZ_DYN_TILE / 256; j++) {
        const uint64_t k = base + (uint64_t)j * 256 + threadIdx.x;
        if (k >= Dn) break;
        const int64_t v = (D.is_nothing && D.is_nothing[k]) ? top : D.values[k];
        switch (w) {
            case 1:  D.out[k] = (uint8_t)v; break;
            case 2:  ((uint16_t *)D.out)[k] = (uint16_t)v; break;
            case 4:  ((uint32_t *)D.out)[k] = (uint32_t)v; break;
            default: ((int64_t *)D.out)[k] = v;
        }
    }
}

// ---- row a3: seg_add_to_local_fixed_do over a column = gather of a field into its context's local -------------------
struct GzdBlob {
    const uint8_t *text; const uint32_t *off, *len; uint32_t n; uint32_t add_nul;
    uint8_t *out; uint64_t *out_len;
    uint64_t *tile;           // scratch [tiles]
    uint32_t pre, pre_len;    // up to 4 bytes in front of every item (little endian in `pre`)
    uint32_t pad_mask, pad_byte;   // pad_to - 1 (0: none): pad_byte's after every item up to the next multiple of pad_to
    uint32_t *item_off, *item_len;   // optional: where every item starts in out; its length with the lead-in
};

// bytes an item of `len` bytes takes in the output
__device__ static inline uint64_t d_blob_item_bytes (const GzdBlob &B, uint32_t len)
{
    const uint64_t raw = (uint64_t)B.pre_len + len + B.add_nul;
    return (raw + B.pad_mask) & ~(uint64_t)B.pad_mask;
}

// grid (tiles, columns)
__global__ void __launch_bounds__(256) k_blob_sum (GzdBlob *cols)
{
    const GzdBlob B = cols[blockIdx.y];
    if (blockIdx.x * 256 >= B.n) return;
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    uint64_t total;
    (void)d_wg_scan_u64 (k < B.n ? d_blob_item_bytes (B, B.len[k]) : 0, threadIdx.x, &total);
    if (!threadIdx.x) B.tile[blockIdx.x] = total;
}

// grid (columns)
__global__ void __launch_bounds__(256) k_blob_scan (GzdBlob *cols)
{
    const GzdBlob B = cols[blockIdx.x];
    const uint64_t total = d_wg_scan_array (B.tile, (B.n + GZ_COL_TILE - 1) / GZ_COL_TILE, threadIdx.x);
    if (!threadIdx.x) *B.out_len = total;
}

#ifndef GZ_BLOB_Q
#define GZ_BLOB_Q 4
#endif
// grid (tiles, columns): the wave copies its 64 snips one after the other, 64 bytes at a time
__global__ void __launch_bounds__(256) k_blob_copy (GzdBlob *cols)
{
    const GzdBlob B = cols[blockIdx.y];
    if (blockIdx.x * 256 >= B.n) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t k = blockIdx.x * 256 + tid;
    const bool on = k < B.n;
    const uint32_t len = on ? B.len[k] : 0, off = (on && len) ? B.off[k] : 0;
    uint64_t total;
    const uint64_t at = B.tile[blockIdx.x] + d_wg_scan_u64 (on ? d_blob_item_bytes (B, len) : 0, tid, &total);
    if (on && B.item_off) B.item_off[k] = (uint32_t)at;
    if (on && B.item_len) B.item_len[k] = len + B.pre_len;
    // the wave's bytes are one contiguous stretch of the output starting at lane 0's offset
    const uint64_t wave_at = ((uint64_t)(uint32_t)__shfl ((int)(uint32_t)(at >> 32), 0) << 32) | (uint32_t)__shfl ((int)(uint32_t)at, 0);
    const uint32_t rel = (uint32_t)(at - wave_at);
    // (The job's fields in registers and the bytes through global pointers: read through a reference into the job table, every field was
    //  fetched again - and waited for - after each store, the text pointer in front of every single load; the kernel was a chain of
    //  trips to the L2 with nothing in flight.)
    const GzGlobalCU8P text = (GzGlobalCU8P)(uintptr_t)B.text; const GzGlobalU8P out = (GzGlobalU8P)(uintptr_t)B.out;
    const uint32_t pre = B.pre, pre_len = B.pre_len, pad_mask = B.pad_mask, pad_byte = B.pad_byte, add_nul = B.add_nul;
    // GZ_BLOB_Q snips per round: their loads are all in flight before the first store (one snip at a time the wave just sat
    // out a memory round trip per snip - 1.1 ms for the 600 MB of SEQ + QUAL of FASTQ-PE-1M)
    uint64_t m = __ballot (on);
    while (m) {
        uint32_t o[GZ_BLOB_Q], l[GZ_BLOB_Q], r[GZ_BLOB_Q], v[GZ_BLOB_Q];
        #pragma unroll
        for (int q = 0; q < GZ_BLOB_Q; q++) {
            const int src = m ? __ffsll ((unsigned long long)m) - 1 : 0;
            const bool live = m != 0;
            m &= m - 1;
            o[q] = (uint32_t)__shfl ((int)off, src); l[q] = live ? (uint32_t)__shfl ((int)len, src) : 0; r[q] = (uint32_t)__shfl ((int)rel, src);
            if (!live) r[q] = 0xffffffffu;
        }
        #pragma unroll
        for (int q = 0; q < GZ_BLOB_Q; q++) v[q] = (uint32_t)lane * 4 < (l[q] & ~3u) ? *(GzGlobalCU32UP)(text + o[q] + lane * 4) : 0;
        // (the 1-3 bytes after a snip's last whole word as well: loaded in the store loop they would cost a trip to memory per snip)
        uint32_t tb[GZ_BLOB_Q];
        #pragma unroll
        for (int q = 0; q < GZ_BLOB_Q; q++) tb[q] = (uint32_t)lane < (l[q] & 3u) ? (uint32_t)text[o[q] + (l[q] & ~3u) + lane] : 0u;
        gz_wait_vector_mem ();                                 // (once, here: not in front of every store, where it would also wait for the stores so far)
        #pragma unroll
        for (int q = 0; q < GZ_BLOB_Q; q++) {
            if (r[q] == 0xffffffffu) continue;
            GzGlobalU8P dst = out + wave_at + r[q];
            GzGlobalCU8P src = text + o[q];
            if ((uint32_t)lane < pre_len) dst[lane] = (uint8_t)(pre >> (8 * lane));
            dst += pre_len;
            if (pad_mask) {                                    // (at most pad_to - 1 bytes)
                const uint32_t raw = pre_len + l[q] + add_nul, padded = (raw + pad_mask) & ~pad_mask;
                if ((uint32_t)lane < padded - raw) dst[l[q] + add_nul + lane] = (uint8_t)pad_byte;
            }
            const uint32_t whole = l[q] & ~3u;
            if ((uint32_t)lane * 4 < whole) *(GzGlobalU32UP)(dst + lane * 4) = v[q];
            for (uint32_t b = 256 + (uint32_t)lane * 4; b < whole; b += 256) *(GzGlobalU32UP)(dst + b) = *(GzGlobalCU32UP)(src + b);
            if ((uint32_t)lane < l[q] - whole) dst[whole + lane] = (uint8_t)tb[q];
            if (add_nul && !lane) dst[l[q]] = 0;
        }
    }
}

// ---- N1 (first part): the line buffer -> lines -> FASTQ records -> tokens -------------------------------------------
// seg_get_next_line (src/seg.c:200-236) for the whole buffer at once: every thread looks at 64 bytes (four 16-byte
// loads), a workgroup at 16 KB; newline counts per workgroup -> scan -> every newline writes the start of the line
// after it; k_lines_finish turns consecutive starts into (offset, length) with the '\r' rule of seg.c:213-216.
#define GZ_NL_PER_THREAD 64
#define GZ_NL_TILE (256 * GZ_NL_PER_THREAD)

struct GzdLines {
    const uint8_t *text; uint64_t n;
    uint32_t *off, *len; uint32_t cap;
    GzLinesResult *result;
    uint32_t *start;          // scratch [cap + 2]: start[j] = offset of line j
    uint64_t *tile;           // scratch [tiles]
    uint32_t byte4;           // the byte looked for, in every byte of the word ('\n': 0x0a0a0a0a)
    uint32_t raw;             // gz_byte_index: just the positions after every occurrence (no "last line without a newline")
};

// bit i set: byte i of the thread's 64 bytes is a newline (bytes beyond n: never)
__device__ static inline uint64_t d_newline_mask (const uint8_t *text, uint64_t n, uint64_t at, uint32_t byte4 = 0x0a0a0a0au)
{
    uint64_t mask = 0;
    if (at + GZ_NL_PER_THREAD <= n) {
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            const gz_u32x4_unaligned v = *(const gz_u32x4_unaligned *)(text + at + 16 * q);
            #pragma unroll
            for (int w = 0; w < 4; w++) {
                const uint32_t t = v[w] ^ byte4;
                const uint32_t z = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu);     // 0x80 in every byte that was '\n'
                const uint32_t bits = ((z >> 7) & 1) | ((z >> 14) & 2) | ((z >> 21) & 4) | ((z >> 28) & 8);
                mask |= (uint64_t)bits << (16 * q + 4 * w);
            }
        }
    }
    else
        for (uint32_t i = 0; at + i < n && i < GZ_NL_PER_THREAD; i++) if (text[at + i] == (byte4 & 0xff)) mask |= 1ull << i;
    return mask;
}

// grid (tiles)
__global__ void __launch_bounds__(256) k_nl_count (GzdLines L)
{
    const uint64_t at = (uint64_t)blockIdx.x * GZ_NL_TILE + (uint64_t)threadIdx.x * GZ_NL_PER_THREAD;
    const uint64_t mask = at < L.n ? d_newline_mask (L.text, L.n, at, L.byte4) : 0;
    uint64_t total;
    (void)d_wg_scan_u64 ((uint64_t)__popcll (mask), threadIdx.x, &total);
    if (!threadIdx.x) L.tile[blockIdx.x] = total;
}

// grid (1)
__global__ void __launch_bounds__(256) k_nl_scan (GzdLines L)
{
    const uint32_t n_tiles = (uint32_t)((L.n + GZ_NL_TILE - 1) / GZ_NL_TILE);
    const uint64_t newlines = d_wg_scan_array (L.tile, n_tiles, threadIdx.x);
    if (threadIdx.x) return;
    const uint64_t lines = newlines + ((!L.raw && L.n && L.text[L.n - 1] != '\n') ? 1 : 0);   // a last line without newline counts (seg.c:227-230)
    L.result->n_lines = lines; L.result->reserved = 0;
    L.result->status = lines <= L.cap ? GZ_ST_OK : GZ_ST_TOO_SMALL;
    L.start[0] = 0;
}

// grid (tiles)
__global__ void __launch_bounds__(256) k_nl_write (GzdLines L)
{
    const uint64_t at = (uint64_t)blockIdx.x * GZ_NL_TILE + (uint64_t)threadIdx.x * GZ_NL_PER_THREAD;
    uint64_t mask = at < L.n ? d_newline_mask (L.text, L.n, at, L.byte4) : 0;
    uint64_t total;
    uint64_t j = L.tile[blockIdx.x] + d_wg_scan_u64 ((uint64_t)__popcll (mask), threadIdx.x, &total);
    for (; mask; mask &= mask - 1, j++)
        if (j + 1 <= (uint64_t)L.cap + 1) L.start[j + 1] = (uint32_t)(at + (uint64_t)(__ffsll ((unsigned long long)mask) - 1) + 1);
}

// grid (tiles of 256 lines over cap)
__global__ void __launch_bounds__(256) k_lines_finish (GzdLines L)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t lines = L.result->n_lines;
    if (j >= lines || j >= L.cap) return;
    const uint64_t start = L.start[j];
    const bool whole = !(j + 1 == lines && L.text[L.n - 1] != '\n');           // ends with a newline
    uint64_t end = whole ? (uint64_t)L.start[j + 1] - 1 : L.n;                  // one past the line's last byte
    if (end > start && L.text[end - 1] == '\r') { end--; L.result->reserved = 1; }   // (any line ended \r\n: result->reserved, see genozip_amd.h)
    L.off[j] = (uint32_t)start;
    L.len[j] = (uint32_t)(end - start);
}

// fastq_seg_get_lines (src/fastq.c:1002-1135) for every read at once. grid (tiles of 256 reads over max_reads)
struct GzdFastq {
    const uint8_t *text; const uint32_t *line_off, *line_len; const GzLinesResult *lines; uint32_t max_reads;
    uint32_t *col[8];         // line 1 (off, len), SEQ, line 3, QUAL
    GzFastqResult *result;
};

__global__ void __launch_bounds__(256) k_fastq_records (GzdFastq F)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t reads = F.lines->n_lines / 4;
    if (reads > F.max_reads) reads = F.max_reads;
    if (!r) { F.result->n_reads = reads; F.result->reserved = 0; }
    if (r >= reads) return;
    const uint32_t *o = F.line_off + 4 * r, *l = F.line_len + 4 * r;
    const bool ok = l[0] >= 1 && F.text[o[0]] == '@' && l[2] >= 1 && F.text[o[2]] == '+' && l[1] == l[3];   // fastq.c:1008-1010,1076,1121
    if (!ok) atomicMin (&F.result->first_bad, (uint32_t)r);
    F.col[0][r] = o[0] + 1; F.col[1][r] = l[0] ? l[0] - 1 : 0;
    F.col[2][r] = o[1];     F.col[3][r] = l[1];
    F.col[4][r] = o[2] + 1; F.col[5][r] = l[2] ? l[2] - 1 : 0;
    F.col[6][r] = o[3];     F.col[7][r] = l[3];
}

// items of a container with known separators (qname_flavors.h:21-49, seg_get_next_item src/seg.c:153-198).
// grid (tiles of 256 snips)
#define GZ_TOK_MAX_SEPS 31
struct GzdTokens {
    const uint8_t *text; const uint32_t *off, *len; uint32_t n;
    uint8_t seps[GZ_TOK_MAX_SEPS + 1]; uint32_t n_seps;
    uint32_t *item_off, *item_len; uint32_t *n_bad;
};

__global__ void __launch_bounds__(256) k_tokenize (GzdTokens T)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= T.n) return;
    const uint32_t len = T.len[k], off = T.off[k];
    const uint8_t *s = T.text + off;                      // (never read when len is 0)
    uint32_t at = 0, i = 0;
    for (; i < T.n_seps; i++) {
        uint32_t e = at;
        const uint8_t sep = T.seps[i];
        while (e < len && s[e] != sep) e++;
        if (e == len) break;
        T.item_off[(uint64_t)i * T.n + k] = off + at; T.item_len[(uint64_t)i * T.n + k] = e - at;
        at = e + 1;
    }
    if (i < T.n_seps) {                                    // a separator is missing: the snip stays whole
        atomicAdd (T.n_bad, 1u);
        for (uint32_t j = 0; j <= T.n_seps; j++) { T.item_off[(uint64_t)j * T.n + k] = off; T.item_len[(uint64_t)j * T.n + k] = j ? 0 : len; }
    }
    else { T.item_off[(uint64_t)T.n_seps * T.n + k] = off + at; T.item_len[(uint64_t)T.n_seps * T.n + k] = len - at; }
}

// seg_integer_or_not over a column (src/seg.c:531-560, str_get_int src/strings.c:315-341): integers (and the context's
// nothing_char) leave the one-character snip SNIP_LOOKUP in the column and go, compacted, to the dyn-int column.
struct GzdIntSplit {
    const uint8_t *text; const uint32_t *off, *len; uint32_t n; uint32_t nothing_char; uint32_t lookup_off;
    uint32_t *snip_off, *snip_len; int64_t *values; uint8_t *is_nothing; uint64_t *n_values;
    uint64_t *tile;           // scratch [tiles]
};

// 0: a snip, 1: an integer (*v), 2: the nothing_char
__device__ static inline int d_int_or_not (const GzdIntSplit &S, uint32_t k, int64_t *v)
{
    const uint32_t len = S.len[k];
    if (!len) return 0;
    const uint8_t *s = S.text + S.off[k];
    if (S.nothing_char && len == 1 && s[0] == (uint8_t)S.nothing_char) { *v = 0; return 2; }
    if ((len == 1 && s[0] == '-') || (len >= 2 && s[0] == '0') || (len >= 2 && s[0] == '-' && s[1] == '0')) return 0;
    const uint32_t negative = s[0] == '-';
    uint64_t out = 0;
    for (uint32_t i = negative; i < len; i++) {
        const uint32_t d = (uint32_t)s[i] - '0';
        if (d > 9 || out > (uint64_t)INT64_MAX / 10) return 0;
        out = out * 10 + d;
        if (out > (uint64_t)INT64_MAX) return 0;
    }
    *v = negative ? -(int64_t)out : (int64_t)out;
    return 1;
}

// grid (tiles)
__global__ void __launch_bounds__(256) k_int_count (GzdIntSplit S)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    int64_t v;
    uint64_t total;
    (void)d_wg_scan_u64 (k < S.n && d_int_or_not (S, k, &v) ? 1 : 0, threadIdx.x, &total);
    if (!threadIdx.x) S.tile[blockIdx.x] = total;
}

// grid (1)
__global__ void __launch_bounds__(256) k_int_scan (GzdIntSplit S)
{
    const uint64_t total = d_wg_scan_array (S.tile, (S.n + 255) / 256, threadIdx.x);
    if (!threadIdx.x) *S.n_values = total;
}

// grid (tiles)
__global__ void __launch_bounds__(256) k_int_write (GzdIntSplit S)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    int64_t v = 0;
    const int kind = k < S.n ? d_int_or_not (S, k, &v) : 0;
    uint64_t total;
    const uint64_t at = S.tile[blockIdx.x] + d_wg_scan_u64 (kind ? 1 : 0, threadIdx.x, &total);
    if (k >= S.n) return;
    if (kind) { S.values[at] = v; S.is_nothing[at] = kind == 2; S.snip_off[k] = S.lookup_off; S.snip_len[k] = 1; }
    else      { S.snip_off[k] = S.off[k]; S.snip_len[k] = S.len[k]; }
}

// ---- row a7, partial case: dyn_int_transpose with a `missing` mask (src/dyn_int.c:64-72,89-96,104-129) --------------
// local holds only the present elements of a rows x cols matrix, in row-major order; the file wants them in
// column-major order. An element's source index is its rank among the present cells in row-major order, its destination
// the rank in column-major order: two prefix sums over the mask (the second over the transposed mask, so that both run
// along memory), then a gather.
struct GzdPartial {
    const uint8_t *missing, *miss_t; uint64_t cells; uint32_t rows, cols, w; uint64_t n_present;
    const uint8_t *in; uint8_t *out;
    uint32_t *rank_a;         // scratch [cells]: row-major rank of every present cell
    uint64_t *tile_a, *tile_b;
    int32_t *status; uint32_t to_file;
};

// grid (tiles of 256 cells, 2): y = 0 the mask, y = 1 the transposed mask
__global__ void __launch_bounds__(256) k_ptr_count (GzdPartial P)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint8_t *m = blockIdx.y ? P.miss_t : P.missing;
    uint64_t total;
    (void)d_wg_scan_u64 (i < P.cells && !m[i] ? 1 : 0, threadIdx.x, &total);
    if (!threadIdx.x) (blockIdx.y ? P.tile_b : P.tile_a)[blockIdx.x] = total;
}

// grid (2)
__global__ void __launch_bounds__(256) k_ptr_scan (GzdPartial P)
{
    const uint64_t total = d_wg_scan_array (blockIdx.x ? P.tile_b : P.tile_a, (uint32_t)((P.cells + 255) / 256), threadIdx.x);
    if (!threadIdx.x && total != P.n_present) *P.status = GZ_ST_CORRUPT;      // the mask and the element count disagree
}

// grid (tiles)
__global__ void __launch_bounds__(256) k_ptr_rank (GzdPartial P)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = i < P.cells && !P.missing[i];
    uint64_t total;
    const uint64_t at = P.tile_a[blockIdx.x] + d_wg_scan_u64 (on ? 1 : 0, threadIdx.x, &total);
    if (on) P.rank_a[i] = (uint32_t)at;
}

// grid (tiles), over the transposed mask: cell i = c * rows + r
__global__ void __launch_bounds__(256) k_ptr_gather (GzdPartial P)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = i < P.cells && !P.miss_t[i];
    uint64_t total;
    const uint64_t d = P.tile_b[blockIdx.x] + d_wg_scan_u64 (on ? 1 : 0, threadIdx.x, &total);
    if (!on || *P.status != GZ_ST_OK) return;
    const uint64_t c = i / P.rows, r = i % P.rows;
    const uint64_t a = P.rank_a[r * P.cols + c];
    const uint64_t src = (P.to_file ? a : d) * P.w, dst = (P.to_file ? d : a) * P.w;
    for (uint32_t k = 0; k < P.w; k++) P.out[dst + k] = P.in[src + k];
}

// gz_device.h -- structures shared by the host orchestrator and the gfx950 kernels of libgenozip_amd.so
//
// Vocabulary (follows the reference): a *stream* is one codec call == the payload of one b250/local section
// (src/codec.h:17-27); a striped codec (RANW/RANw/ARTW/ARTw) splits it into 4 byte *planes*
// (src/htscodecs/rANS_static4x16pr.c:1165-1227, arith_dynamic.c:636-753) and tries several methods per plane; each
// (plane, method) pair - or the whole stream for the unstriped codecs - is a *leaf*: one independent entropy-coding
// job. Leaves are the unit of GPU parallelism: the coders are serial per leaf, so throughput comes from running
// thousands of leaves (VBlocks x contexts x planes x methods) at once.
#pragma once
#include <stdint.h>

#define GZ_X_PACK   0x80
#define GZ_X_RLE    0x40
#define GZ_X_CAT    0x20
#define GZ_X_NOSZ   0x10
#define GZ_X_STRIPE 0x08

#define GZ_ENG_NONE  0
#define GZ_ENG_RANS  1
#define GZ_ENG_ARITH 2

#define GZ_TAB_CAP     (257 * 257 * 3 + 1024)   // order-1 frequency table bytes (rans_compress_bound_4x16)
#define GZ_ROW_SLOT    768                      // serialised bytes of one order-1 row: <= 256 * (2 + zero-run)
#define GZ_PREFIX_CAP  32

// encoder-side description of one (context, symbol): x' = x + bias + ((x * rcp) >> rsh) * cmpl
// (src/htscodecs/rANS_word.h:169-265; 16 bytes instead of the reference's 24)
struct GzRansSym { uint32_t x_max, rcp, bias, cmpl_rsh; };   // cmpl_rsh = cmpl | (rsh - 32) << 16

struct GzdStream {
    const uint8_t  *in;
    const uint32_t *in_len_dev;
    uint8_t  *out;            // payload destination (batch mode) / scratch (vb mode: final copy goes to z_data)
    uint8_t  *planes;         // n bytes, striped codecs only
    uint32_t in_len;          // planning upper bound
    uint32_t out_cap;
    uint32_t n;               // resolved length
    uint32_t out_len;         // result: payload bytes
    int32_t  status;
    int32_t  codec_req;
    uint32_t first_leaf, n_leaves;
    uint8_t  codec;           // effective codec (< 50 B -> NONE in section mode)
    uint8_t  engine, order, striped;
    uint8_t  best_leaf[4];    // index (relative to first_leaf) of the winning leaf of each plane
    uint8_t  whole_leaf;      // index (relative to first_leaf) of the leaf that codes the unstriped stream
    uint32_t plane_unit_len[4];
    // section / VBlock mode
    int32_t  vb;              // -1 in batch mode
    uint32_t sec_in_vb;
    uint64_t z_off;           // where the 40-byte header of this section starts inside the VB's z_data
    uint8_t  hdr[40];         // SectionHeaderCtx template; lengths, codec and digest are patched on device
    uint32_t *out_len_dev;    // batch mode, optional: the payload length for a later section writer
    uint32_t raw_len;         // precompressed section: data_uncompressed_len of the header
    uint8_t  pre;             // precompressed section: `in` already is the payload of codec hdr[25] (hdr[26] when hdr_codec)
    uint8_t  hdr_codec;       // a complex codec (DOMQ) names the section: the coder of the stream goes to sub_codec
    uint32_t emit_a, emit_w, emit_done;   // k_emit over several workgroups: the slices' adler32 sums (mod 65521) and how many are through
};

struct GzdLeaf {
    uint32_t stream;
    uint8_t  plane;           // 0..3, or 0xff = whole stream
    uint8_t  method;          // htscodecs order byte of this leaf (without STRIPE)
    uint8_t  engine;
    // resolved by k_leaf_prep:
    uint8_t  active;
    uint8_t  flag;            // final first byte of the unit
    uint8_t  o1, rle, packed_on, cat;
    uint8_t  prefix_len;
    uint8_t  shift_bits;      // rANS order-1: 10 or 12
    uint8_t  tile_models;     // arith: the host allows k_arith_model_tiled for this leaf (the device decides by alphabet and order: d_leaf_tiled)
    uint8_t  prefix[GZ_PREFIX_CAP];    // flag [varint n] [pack meta] [varint packed n]
    const uint8_t *src;  uint32_t n;
    const uint8_t *coded; uint32_t coded_n;   // bytes the entropy coder sees (== packed or src)
    uint32_t max_sym;         // arith: 1 + largest byte
    uint32_t nsym;            // distinct coded byte values
    // (two small device-only tables, written by k_leaf_prep: kept out of this struct, which is uploaded for every leaf)
    uint8_t  *symlist;        // [256] the distinct coded byte values, ascending
    uint16_t *symrank;        // [256] value -> rank, 0xffff if absent (rank 255 is a legal rank)
    uint32_t tab_len, pay_len, unit_len;
    int32_t  overflow;        // payload alone already > coded_n => CAT
    // scratch owned by this leaf (device pointers; NULL when the method does not need it)
    uint8_t   *packed;        // coded bytes after PACK (n + 1)
    uint32_t  *F;             // histogram: 256 (order 0) or 256*256 (order 1) counters, + 256 row totals
    GzRansSym *syms;          // 256 or 256*256 records
    uint8_t   *rowbuf;        // order-1: GZ_ROW_SLOT bytes per context while serialising; also nested table coder scratch
    uint8_t   *tab;           // serialised frequency table
    uint8_t   *pay;           // entropy-coded payload area (rANS fills it from the end)
    uint32_t  *models;        // arith (run-length variant): global-memory models when they do not fit the LDS
    uint8_t   *triples;       // arith: 12 bytes per coded byte: 2^-45 / tot as a double with cum in its low 16 bits, freq as the high word of the double freq * 2^45  (k_arith_model -> k_arith_chain, k_chain_expand)
    uint32_t  *spos;          // arith order-1: positions grouped by context (the byte before), stream order inside a context
    uint8_t   *srk;           // arith order-1: static rank of the symbol at spos[j]; NULL (leaves under 2^24 positions): spos[j] = position << 8 | rank
    uint32_t  *ctxoff;        // arith order-1: [tile][context] -> index into spos/srk of the first occurrence at or after the tile
    uint32_t  *ctxend;        // arith order-1: [position chunk][context] -> end of the context's run in that chunk's part of the sorted lists
    uint16_t  *ev_ctx;        // arith run-length variant: model id of every coding event (0..255 literal models, 256 + 0..257 run models)
    uint8_t   *ev_sym;        // arith run-length variant: its symbol (literal: rank in the leaf's alphabet; run digit: 0..3)
    uint32_t  arith_n;        // arith: symbols the coder sees: coded_n, or the number of events of the run-length variant
    uint32_t  nctx;           // arith: row length of ctxoff / ctxend: 256, or 768 for the run-length variant's 514 models
    uint32_t  *mstate;        // arith: the models' registers between two position chunks (GZ_MSTATE_WORDS x 64 lanes per context)
    uint32_t   pres[8];       // which byte values occur in the leaf's source bytes (k_presence, many workgroups; k_leaf_prep is one)
    uint64_t  *succ;          // arith, order 1, leaves that span position chunks: 256 rows x 4 words - which symbols (leaf ranks) follow each context byte anywhere in the leaf
    uint8_t   *events;        // arith: one 32-bit digit per output byte (k_low_replay / k_low_norm)
    uint8_t   *rvals;         // arith: a = cum * r of every symbol, what it adds to low (k_chain_expand -> k_low_scatter)
    uint8_t   *kbits;         // arith: per 64-symbol slice 16 bytes: two bits per symbol, the bytes low moves up after it (k_chain_expand -> k_low_scatter)
    uint8_t   *ckpt;          // arith: the chain's state (range * 2^-7 as a double) before every 64th symbol and after the last (k_arith_chain -> k_chain_expand)
    uint8_t   *kpos;          // arith: per 64-symbol slice: shifts in it, then (k_low_scan) shifts before it
    uint8_t   *resid;         // arith: per slice: what is left in its 32-bit window (+ carry) after its last shift
    uint32_t  n_events;
    uint32_t  low_base;       // arith: shifts in the position chunks the low kernels have been through
    uint32_t  touch_sink;     // k_arith_chain: blocks of 64 symbols that went the slow way (a total below 256 in them) - diagnostics
    uint32_t  pay_cap;
};

struct GzdVB {
    uint8_t *z_data; uint64_t z_cap; uint64_t z_len;
    uint32_t first_stream, n_streams;
    uint32_t vblock_i, recon_size, longest_line_len, longest_seq_len;
    uint8_t  digest[16];
    uint8_t  vb_flags;
    int32_t  status;
    uint32_t mark_stream, mark_index;  // in: a position in the VBlock's stream list; out: how many sections were written in front of it
};

// ---- decode side ----
struct GzdDecStream {
    const uint8_t  *in; uint32_t in_len;
    uint8_t  *out; uint32_t out_len;   // exact expected length
    uint8_t  *tmp_planes;              // out_len bytes: plane buffer (striped)
    uint8_t  *tmp_packed;              // out_len bytes per leaf slot x4: packed intermediate
    int32_t  codec, status;
    uint32_t first_leaf;               // 4 decode leaves reserved per stream
    uint8_t  striped, engine, n_leaves, pad;
};

struct GzdDecLeaf {
    uint32_t stream;
    uint8_t  active, engine, o1, rle, cat, packed_on, per, shift_bits;
    uint8_t  map[16];
    const uint8_t *body; uint32_t body_len;   // after the unit's prefix
    uint8_t  *dst; uint32_t coded_n;          // entropy decoder output (packed tmp or final plane / out)
    uint8_t  *final_dst; uint32_t n;          // after unpack
    int32_t  status;
    // scratch
    uint8_t  *lut;        // rANS: o0: 4096 * 4 B {sym,freq,off}; o1: 256 << bits sym bytes
    uint32_t *fc;         // rANS o1: 256*256 {freq | cum << 16}
    uint8_t  *tabtmp;     // uncompressed order-1 table
    uint32_t *models;
    const uint8_t *pay; uint32_t pay_len;     // start of the 4 states
};

// gz_merge.h -- row a4 of SURVEY.md 8(a): the ordered dictionary merge, HOST code (no kernels; included by gz_host.cpp).
//
// Reference: ctx_merge_in_one_vctx src/context.c:938-1079, ctx_commit_node :269-316, hash_global_get_entry
// src/hash.c:444-482 with the singleton tables :280-366, ctx_insert_to_dict src/context.c:50-71, add_count :925-934,
// ctx_drop_all_the_same :795-871, hash_next_size_up src/hash.c:26-48.
//
// The reference keeps, per file-level context (zctx): the dictionary (snips separated by NUL, in word-index order), a node
// per word, a chained hash over a prime-sized table and - for snips seen exactly once - a table of CRC32C digests
// ("singletons") whose text went to the VBlock's `local` instead of the dictionary. What reaches the file is: the word
// index every VBlock node gets (-> b250), the dictionary order, the counts, which snips were diverted to local, and
// whether an all-the-same b250 may be dropped. This file computes exactly that with structures of its own: an
// open-addressing table over the 64-bit snip mix for the words (lookups are exact string matches, so the chain layout of
// the reference is unobservable) and a counter per (bucket, digest) for the singletons - the reference's per-bucket linked
// list is searched by digest only, so equal (bucket, digest) entries are indistinguishable and a count says it all; the
// bucket is the reference's: hash_do (snip) mod the prime chosen when the context was first merged.
//
// Serial per context by nature (word indices are handed out in arrival order): contexts are independent of each other, so
// the caller may merge different contexts on different host threads; one context must see its VBlocks in order.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include <unordered_map>
#include "../../include/genozip_amd.h"

struct GzZctx {
    std::vector<uint8_t>  dict;           // zctx->dict
    std::vector<uint64_t> char_index;     // CtxNode.char_index per word
    std::vector<uint32_t> snip_len;       // CtxNode.snip_len per word
    std::vector<uint64_t> counts;         // zctx->counts
    std::vector<uint64_t> mix;            // the 64-bit rotate-xor value of every word (own table's key)
    std::vector<uint32_t> slots;          // open addressing: word index or 0xffffffff
    uint32_t slot_bits = 0;
    uint32_t hash_len = 0;                // zctx->global_hash.len (a prime, hash.c:34-40)
    std::unordered_map<uint64_t, uint32_t> stons;   // (bucket << 32 | crc32c) -> live singletons with that key
    uint64_t n_stons = 0, n_failed_stons = 0;
    uint8_t  flags = 0; bool flags_set = false;      // zctx->flags: the flags of vb_i=1 (context.c:962-963)
    int32_t  all_the_same_wi = -1;                   // zctx->dict_flags.all_the_same_wi once set
    bool     rm_dict_all_the_same = false, override_rm_dict_ats = false;
    uint8_t  lcodec = 0, bcodec = 0;                 // committed by codec_assign_best_codec (codec.c:352-363)
    uint32_t num_new_entries_prev_merged_vb = 0;
};

static const uint32_t GZ_NO_WORD = 0xffffffffu;

// hash_do's 64-bit mix before the modulo (src/hash.h:30-52)
static inline uint64_t gz_snip_mix (const uint8_t *s, uint32_t n)
{
    uint64_t r = 0;
    for (uint32_t i = 0; i < n; i++) r = ((r << 23) | (r >> 41)) ^ (uint64_t)s[i];
    return r;
}

// hash_crc32 (src/hash.c:241-272): CRC-32C (Castagnoli), initial value 0, no final xor - what chaining the SSE4.2
// crc32 instructions over the snip gives (a 64/32/16-bit step equals that many byte steps)
static inline uint32_t gz_crc32c (const uint8_t *s, uint32_t n)
{
    static uint32_t T[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            T[i] = c;
        }
        init = true;
    }
    uint32_t crc = 0;
    for (uint32_t i = 0; i < n; i++) crc = T[(crc ^ s[i]) & 0xff] ^ (crc >> 8);
    return crc;
}

extern "C" uint32_t gz_hash_next_size_up (uint64_t size)
{
    // src/hash.c:26-48 (segconf.vb_size at its 16 MB floor for the cap; allow_huge = false)
    static const uint32_t sizes[] = { 65521, 92681, 131071, 185363, 262139, 370723, 524287, 741431, 1048573, 1482907, 2097143,
                                      2965819, 4194301, 5931641, 8388593, 11863279, 16777213, 19951579, 23726561, 28215799,
                                      33554393, 39903161, 47453111, 56431601, 67108859 };
    if (size > 16000000) size = 16000000;
    for (uint32_t s : sizes) if (size < s) return s;
    return sizes[sizeof (sizes) / sizeof (sizes[0]) - 1];
}

extern "C" GzZctx *gz_zctx_create (uint32_t estimated_entries)
{
    GzZctx *z = new GzZctx ();
    if (!estimated_entries) estimated_entries = 1000;                     // hash.c:229
    z->hash_len = gz_hash_next_size_up ((uint64_t)estimated_entries * 3); // hash.c:231
    z->slot_bits = 10;
    z->slots.assign ((size_t)1 << z->slot_bits, GZ_NO_WORD);
    return z;
}

extern "C" void gz_zctx_destroy (GzZctx *z) { delete z; }

static inline size_t zctx_slot (const GzZctx *z, uint64_t mix) { return (size_t)((mix * 0x9E3779B97F4A7C15ull) >> (64 - z->slot_bits)); }

static uint32_t zctx_find (const GzZctx *z, uint64_t mix, const uint8_t *s, uint32_t n)
{
    const size_t mask = z->slots.size () - 1;
    for (size_t p = zctx_slot (z, mix);; p = (p + 1) & mask) {
        const uint32_t w = z->slots[p];
        if (w == GZ_NO_WORD) return GZ_NO_WORD;
        if (z->mix[w] == mix && z->snip_len[w] == n && !memcmp (z->dict.data () + z->char_index[w], s, n)) return w;
    }
}

static void zctx_table_put (GzZctx *z, uint32_t w)
{
    const size_t mask = z->slots.size () - 1;
    size_t p = zctx_slot (z, z->mix[w]);
    while (z->slots[p] != GZ_NO_WORD) p = (p + 1) & mask;
    z->slots[p] = w;
}

static uint32_t zctx_add_word (GzZctx *z, uint64_t mix, const uint8_t *s, uint32_t n)
{
    const uint32_t w = (uint32_t)z->snip_len.size ();
    z->char_index.push_back (z->dict.size ());             // ctx_insert_to_dict: the snip and a NUL
    z->dict.insert (z->dict.end (), s, s + n);
    z->dict.push_back (0);
    z->snip_len.push_back (n);
    z->counts.push_back (0);
    z->mix.push_back (mix);
    if (2 * (size_t)(w + 1) > z->slots.size ()) {            // keep the table at most half full
        z->slot_bits++;
        z->slots.assign ((size_t)1 << z->slot_bits, GZ_NO_WORD);
        for (uint32_t k = 0; k <= w; k++) zctx_table_put (z, k);
    }
    else zctx_table_put (z, w);
    return w;
}

// ctx_commit_node. ston_out: where a snip that turns out to be a (new) global singleton goes (seg_add_to_local_fixed_do
// with add_nul, context.c:297); returns the word index (of the SNIP_LOOKUP word for such a snip), < 0 on overflow
static int64_t zctx_commit (GzZctx *z, const uint8_t *s, uint32_t n, bool allow_singleton, uint8_t *ston_out, uint64_t ston_cap, uint64_t *ston_len, uint32_t *n_stons)
{
    const uint64_t mix = gz_snip_mix (s, n);
    const uint32_t w = zctx_find (z, mix, s, n);
    if (w != GZ_NO_WORD) return w;                                         // existing node (hash.c:479-481)
    const uint64_t key = ((uint64_t)(mix % z->hash_len) << 32) | gz_crc32c (s, n);
    bool was_ston = false;                                                 // hash_stons_remove_singleton (hash.c:280-326)
    if (z->n_stons) {
        auto it = z->stons.find (key);
        if (it != z->stons.end ()) {
            was_ston = true;
            if (!--it->second) z->stons.erase (it);
            z->n_stons--; z->n_failed_stons++;
        }
    }
    if (!was_ston && allow_singleton) {                                    // a new singleton (hash.c:461-465)
        z->stons[key]++; z->n_stons++;
        if (*ston_len + n + 1 > ston_cap) return -1;
        if (n) memcpy (ston_out + *ston_len, s, n);
        ston_out[*ston_len + n] = 0;
        *ston_len += (uint64_t)n + 1;
        ++*n_stons;
        static const uint8_t lookup[1] = { 1 };                            // SNIP_LOOKUP (context.h:34)
        return zctx_commit (z, lookup, 1, false, ston_out, ston_cap, ston_len, n_stons);
    }
    return zctx_add_word (z, mix, s, n);                                   // a new node (hash.c:468-475)
}

// add_count (context.c:925-934): bit 31 of a VBlock count = "protected from removal", carried to bit 63
static inline void gz_add_count (uint64_t *counter, uint32_t inc)
{
    if (inc & 0x80000000u) { *counter += inc & 0x7fffffffu; *counter |= 0x8000000000000000ull; }
    else *counter += inc;
}

extern "C" int gz_ctx_merge (GzZctx *z, GzMergeJob *j)
{
    if (!z || !j || j->n_ol > z->snip_len.size () || (j->n_new && (!j->dict || !j->node_char_index || !j->node_snip_len || !j->node2word)) ||
        ((j->n_new || j->n_ol) && !j->counts)) return GZ_ERR_ARG;
    j->ston_len = 0; j->n_stons = 0; j->dropped_b250 = 0;
    z->num_new_entries_prev_merged_vb = j->n_new;
    if (j->vblock_i == 1 && (j->b250_len || j->local_len)) { z->flags = j->flags; z->flags_set = true; }   // context.c:962-963
    if (!j->lcodec) j->lcodec = z->lcodec;                                                                 // context.c:980-981
    if (!j->bcodec) j->bcodec = z->bcodec;
    const bool can_ston = j->can_have_singletons != 0;
    for (uint32_t i = 0; i < j->n_new; i++) {
        const uint32_t count = j->counts[j->n_ol + i];
        const int64_t wi = zctx_commit (z, j->dict + j->node_char_index[i], j->node_snip_len[i], count == 1 && can_ston,
                                        j->ston_local, j->ston_local ? j->ston_cap : 0, &j->ston_len, &j->n_stons);
        if (wi < 0) return GZ_TOO_SMALL;                                   // ston_local too small (cap >= the VBlock's dict length always suffices)
        gz_add_count (&z->counts[wi], count);
        j->node2word[i] = (int32_t)wi;                                     // context.c:1032
    }
    for (uint32_t ni = 0; ni < j->n_ol; ni++) gz_add_count (&z->counts[ni], j->counts[ni]);   // context.c:1059-1060

    // ctx_drop_all_the_same (context.c:795-871)
    const uint8_t ATS = 1u << 5;                                           // FlagsCtx.all_the_same (sections.h:99-118)
    if (!(j->flags & ATS)) { z->override_rm_dict_ats = true; return GZ_OK; }
    bool drop = !j->no_drop_b250;
    if (drop && j->pair2_identical && (j->b250_r1_len || (j->local_r1_len && !(j->local_len + j->ston_len)))) drop = false;
    int64_t wi = -1;
    if (drop) {
        const int32_t ni = j->ats_node_index;                              // the only b250 entry of the context
        wi = ni < 0 ? ni : (uint32_t)ni < j->n_ol ? ni : (uint32_t)ni - j->n_ol < j->n_new ? j->node2word[(uint32_t)ni - j->n_ol] : -1;
        if (wi < 0 || wi > 15) drop = false;                               // MAX_ALL_THE_SAME_WI (sections.h:123)
    }
    bool simple_lookup = false;
    if (drop) {
        const uint8_t *snip = z->dict.data () + z->char_index[wi];
        if (snip[0] == 5) drop = false;                                    // SNIP_SELF_DELTA
        simple_lookup = snip[0] == 1 && !snip[1];                          // SNIP_LOOKUP alone
        if (drop && (j->local_len + j->ston_len) && !simple_lookup) drop = false;
    }
    if (drop) {
        const uint8_t mine = j->flags & ~ATS, vb1 = (j->vblock_i == 1 ? 0 : z->flags) & ~ATS;
        if (mine != vb1) drop = false;
    }
    if (drop) {
        if (z->all_the_same_wi < 0) z->all_the_same_wi = (int32_t)wi;
        else if (z->all_the_same_wi != wi) drop = false;
    }
    if (!drop) { z->override_rm_dict_ats = true; return GZ_OK; }
    j->dropped_b250 = 1;
    if (simple_lookup) z->rm_dict_all_the_same = true;
    return GZ_OK;
}

extern "C" int gz_zctx_view (const GzZctx *z, GzZctxView *v)
{
    if (!z || !v) return GZ_ERR_ARG;
    v->dict = z->dict.data (); v->dict_len = z->dict.size ();
    v->char_index = z->char_index.data (); v->snip_len = z->snip_len.data (); v->counts = z->counts.data ();
    v->n_words = (uint32_t)z->snip_len.size (); v->hash_len = z->hash_len;
    v->n_singletons = z->n_stons; v->n_failed_singletons = z->n_failed_stons;
    v->flags = z->flags; v->all_the_same_wi = z->all_the_same_wi;
    v->rm_dict_all_the_same = z->rm_dict_all_the_same && !z->override_rm_dict_ats;
    v->lcodec = z->lcodec; v->bcodec = z->bcodec;
    return GZ_OK;
}

// codec_assign_best_codec's commit to the file-level context (codec.c:352-363)
extern "C" int gz_zctx_commit_codec (GzZctx *z, int is_local, int codec)
{
    if (!z) return GZ_ERR_ARG;
    (is_local ? z->lcodec : z->bcodec) = (uint8_t)codec;
    return GZ_OK;
}

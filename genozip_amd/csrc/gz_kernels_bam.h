// gz_kernels_bam.h -- N1 for BAM (SURVEY 8(0) row configs[2]; bam_seg_txt_line, src/bam_seg.c:425-520): the alignment records of an
// uncompressed BAM stream (what is left after the BGZF layer, which is I/O) -> the fields the SAM plan segs.
//
// The reference walks a VBlock's records one after the other - block_size, the fixed fields, read_name, then the three conversions
// bam_seq_to_sam (src/bam_seq.c:58-103: two bases per byte through "=ACMGRSVTWYHKDBN"), sam_cigar_binary_to_textual
// (src/sam_cigar.c:155-206: length then "MIDNSHP=X") and bam_rewrite_qual (src/bam_seg.c:276-284: + 33, 0xff = missing) - and segs
// the textual forms with the functions SAM uses. Here the same in two data-parallel steps:
//
//   gz_bam_records   where the records start. The chain (every record's block_size names the next record, bam_unconsumed_scan_forwards
//                    src/bam_seg.c:49-67) is a linked list through memory - one dependent trip to memory per record if walked by one
//                    thread. So the stream is cut into chunks of 64 KB: every chunk GUESSES where its first record starts - the first
//                    position that looks like an alignment by the tests the reference itself uses to find one in the middle of a
//                    stream (bam_unconsumed_scan_backwards, src/bam_seg.c:76-130: block_size against l_read_name / n_cigar_op / l_seq,
//                    a NUL-terminated printable read_name, pos >= -1) and whose successor does too -, all chunks walk their stretch
//                    at once, and every chunk checks its guess against where the chunk before it ended: if they all agree they are
//                    all right (chunk 0 is right by definition and each vouches for the next); otherwise one thread goes through
//                    the chunks from the first wrong one and walks those again that do not start where their predecessor ended
//                    (rare: the tests are strict). The records found are exactly those of the serial walk from position 0.
//   gz_bam_to_sam    every record's alignment line: lengths (a thread per record), a prefix sum, then the text - the fixed fields by
//                    the record's thread, SEQ / QUAL (two thirds of the line) by all lanes of the wave, one record after the other.
#pragma once
#include "gz_device.h"
#include "gz_devutil.h"
#include "gz_kernels_seg.h"

#define GZ_BAM_CHUNK 65536u
#define GZ_BAM_NONE  0xffffffffu

struct GzdBamChain {
    const uint8_t *bam; uint64_t n;
    uint32_t *rec_off; uint32_t cap; GzBamResult *result;
    uint32_t *entry, *exit_, *count;   // scratch [chunks + 1]
    uint64_t *tile;                    // scratch [chunks + 1]: records before every chunk
    int32_t n_ref;
};

__device__ static inline uint32_t d_le32 (const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
__device__ static inline uint32_t d_le16 (const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }

// what any record must satisfy for the walk to go on (bam_seg.c:444-447): the fixed part fits, the record ends inside the stream
__device__ static inline bool d_bam_walkable (const uint8_t *bam, uint64_t n, uint64_t p, uint32_t *block_size)
{
    if (p + 36 > n) return false;
    const uint32_t bs = d_le32 (bam + p);
    *block_size = bs;
    return bs >= 32 && (uint64_t)bs + 4 <= n - p;
}

// does position p look like the start of an alignment (bam_unconsumed_scan_backwards, bam_seg.c:76-100; + the reference ids the
// segmenter insists on, :459-460)
__device__ static inline bool d_bam_plausible (const uint8_t *bam, uint64_t n, uint64_t p, int32_t n_ref, uint32_t *block_size)
{
    uint32_t bs;
    if (!d_bam_walkable (bam, n, p, &bs)) return false;
    *block_size = bs;
    if (bs > 100000000u) return false;
    const uint8_t *a = bam + p;
    const int32_t ref_id = (int32_t)d_le32 (a + 4), pos = (int32_t)d_le32 (a + 8), next_ref = (int32_t)d_le32 (a + 24), next_pos = (int32_t)d_le32 (a + 28);
    const uint32_t l_read_name = a[12], n_cigar = d_le16 (a + 16), l_seq = d_le32 (a + 20);
    if (l_seq > bs) return false;
    if ((uint64_t)32 + 4ull * n_cigar + l_read_name + l_seq + (l_seq + 1) / 2 > bs) return false;
    if (l_read_name < 2) return false;
    if (pos < -1 || next_pos < -1 || ref_id < -1 || ref_id >= n_ref || next_ref < -1 || next_ref >= n_ref) return false;
    if (a[36 + l_read_name - 1] != 0) return false;
    for (uint32_t i = 0; i + 1 < l_read_name; i++) if (a[36 + i] < '!' || a[36 + i] > '~') return false;
    return true;
}

// grid (chunks), 64 threads: entry[c] = the first position in chunk c that looks like an alignment and is followed by one (or by the end)
__global__ void __launch_bounds__(64) k_bam_entry (GzdBamChain B)
{
    const uint32_t c = blockIdx.x;
    const int lane = threadIdx.x;
    if (!c) { if (!lane) B.entry[0] = B.n ? 0 : GZ_BAM_NONE; return; }
    const uint64_t p0 = (uint64_t)c * GZ_BAM_CHUNK, p1 = p0 + GZ_BAM_CHUNK < B.n ? p0 + GZ_BAM_CHUNK : B.n;
    uint32_t found = GZ_BAM_NONE;
    for (uint64_t base = p0; base < p1 && found == GZ_BAM_NONE; base += 64) {
        const uint64_t p = base + lane;
        bool ok = false;
        uint32_t bs = 0, bs2;
        if (p < p1 && d_bam_plausible (B.bam, B.n, p, B.n_ref, &bs)) {
            const uint64_t q = p + 4 + bs;
            ok = q == B.n || d_bam_plausible (B.bam, B.n, q, B.n_ref, &bs2);
        }
        const uint64_t m = __ballot (ok);
        if (m) found = (uint32_t)(base - p0) + (uint32_t)(__ffsll ((unsigned long long)m) - 1);
    }
    if (!lane) B.entry[c] = found;          // (relative to the chunk's start)
}

// the records of chunk c from `at` on (relative to the chunk's start; GZ_BAM_NONE: no record starts in it): how many, and where
// the first record beyond the chunk starts (absolute). Returns false if a record is not walkable.
__device__ static inline bool d_bam_walk (const GzdBamChain &B, uint32_t c, uint32_t at, uint32_t *count, uint64_t *end, uint32_t *out)
{
    const uint64_t p0 = (uint64_t)c * GZ_BAM_CHUNK, p1 = p0 + GZ_BAM_CHUNK < B.n ? p0 + GZ_BAM_CHUNK : B.n;
    uint32_t k = 0;
    uint64_t p = p0 + at;
    if (at == GZ_BAM_NONE) { *count = 0; *end = 0; return true; }
    while (p < p1) {
        uint32_t bs;
        if (!d_bam_walkable (B.bam, B.n, p, &bs)) { *count = k; *end = p; return false; }
        if (out) out[k] = (uint32_t)p;
        k++;
        p += 4 + (uint64_t)bs;
    }
    *count = k; *end = p;
    return true;
}

// grid (ceil (chunks / 64)), 64 threads: a thread per chunk
__global__ void __launch_bounds__(64) k_bam_walk_count (GzdBamChain B, uint32_t n_chunks)
{
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_chunks) return;
    uint32_t k; uint64_t end;
    const bool ok = d_bam_walk (B, c, B.entry[c], &k, &end, NULL);
    B.count[c] = ok ? k : GZ_BAM_NONE;
    B.exit_[c] = (uint32_t)end;             // (streams are < 4 GB)
}

// grid (ceil (chunks / 64)), 64 threads: is chunk c's guess where the chunk before it (the last one a record starts in) ended?
// If every chunk says yes, all of them are right (chunk 0 is right by definition, and each vouches for the next).
__global__ void __launch_bounds__(64) k_bam_check (GzdBamChain B, uint32_t n_chunks, uint32_t *first_wrong)
{
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_chunks) return;
    bool ok = B.count[c] != GZ_BAM_NONE;
    if (c && ok) {
        uint32_t b = c - 1;
        while (b && B.entry[b] == GZ_BAM_NONE) b--;            // (records longer than a chunk: rare)
        const uint64_t next = B.exit_[b], p0 = (uint64_t)c * GZ_BAM_CHUNK, p1 = p0 + GZ_BAM_CHUNK < B.n ? p0 + GZ_BAM_CHUNK : B.n;
        const uint32_t want = (next >= p0 && next < p1) ? (uint32_t)(next - p0) : GZ_BAM_NONE;
        ok = B.entry[b] != GZ_BAM_NONE && next >= p0 && B.entry[c] == want;
    }
    if (!ok) atomicMin (first_wrong, c);
}

// one workgroup. All guesses right (the usual case): the records before every chunk by a prefix sum. Otherwise thread 0 goes through
// the chunks in order from the first wrong one: a chunk that does not start where the one before it ended is walked again.
__global__ void __launch_bounds__(256) k_bam_fix (GzdBamChain B, uint32_t n_chunks, const uint32_t *first_wrong)
{
    uint32_t *sh = (uint32_t *)(gz_lds + 4096);
    const uint32_t fw = *first_wrong;
    if (fw != GZ_BAM_NONE && !threadIdx.x) {
        uint64_t next = 0;
        uint32_t rewalked = 0, bad = GZ_BAM_NONE;
        int32_t status = GZ_ST_OK;
        for (uint32_t b = 0; b < fw; b++) if (B.entry[b] != GZ_BAM_NONE) next = B.exit_[b];
        uint64_t before = 0;
        for (uint32_t c = fw; c < n_chunks && status == GZ_ST_OK; c++) {
            const uint64_t p0 = (uint64_t)c * GZ_BAM_CHUNK, p1 = p0 + GZ_BAM_CHUNK < B.n ? p0 + GZ_BAM_CHUNK : B.n;
            const uint32_t want = next < p1 ? (uint32_t)(next - p0) : GZ_BAM_NONE;
            if (B.entry[c] != want || B.count[c] == GZ_BAM_NONE) {
                uint32_t k; uint64_t end;
                B.entry[c] = want;
                const bool ok = d_bam_walk (B, c, want, &k, &end, NULL);
                B.count[c] = k; B.exit_[c] = (uint32_t)end;
                rewalked++;
                if (!ok) { status = GZ_ST_CORRUPT; for (uint32_t b = 0; b < c; b++) before += B.count[b]; bad = (uint32_t)(before + k); }
            }
            if (B.entry[c] != GZ_BAM_NONE) next = B.exit_[c];
        }
        sh[0] = (uint32_t)status; sh[1] = bad; sh[2] = rewalked;
    }
    else if (!threadIdx.x) { sh[0] = GZ_ST_OK; sh[1] = GZ_BAM_NONE; sh[2] = 0; }
    __syncthreads ();
    int32_t status = (int32_t)sh[0];
    for (uint32_t c = threadIdx.x; c < n_chunks; c += 256) B.tile[c] = status == GZ_ST_OK ? B.count[c] : 0;
    __syncthreads ();
    const uint64_t total = d_wg_scan_array (B.tile, n_chunks, threadIdx.x);
    if (!threadIdx.x) {
        uint32_t bad = sh[1];
        if (status == GZ_ST_OK) {                               // the last record must end where the stream ends (bam_seg.c:62-66)
            uint64_t next = 0;
            for (uint32_t b = n_chunks; b--; ) if (B.entry[b] != GZ_BAM_NONE) { next = B.exit_[b]; break; }
            if (next != B.n) { status = GZ_ST_CORRUPT; bad = (uint32_t)total; }
        }
        if (status == GZ_ST_OK && total > B.cap) status = GZ_ST_TOO_SMALL;
        B.result->n_records = status == GZ_ST_CORRUPT ? 0 : total; B.result->text_len = 0; B.result->status = status; B.result->first_bad = bad; B.result->n_rewalked = sh[2];
    }
}

// grid (ceil (chunks / 64)), 64 threads
__global__ void __launch_bounds__(64) k_bam_walk_write (GzdBamChain B, uint32_t n_chunks)
{
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_chunks || B.result->status != GZ_ST_OK) return;
    uint32_t k; uint64_t end;
    (void)d_bam_walk (B, c, B.entry[c], &k, &end, B.rec_off + B.tile[c]);
}

// ---- records -> alignment lines ---------------------------------------------------------------------------------------------
struct GzdBamText {
    const uint8_t *bam; uint64_t n; const uint32_t *rec_off; uint32_t n_rec;
    const uint8_t *ref_names; const uint32_t *ref_name_off; int32_t n_ref;
    uint8_t *text; uint64_t text_cap; uint32_t *line_off; GzBamResult *result;
    uint32_t *len;                     // scratch [n_rec]
    uint64_t *tile;                    // scratch [tiles]
};

__device__ static inline uint32_t d_dec_len (uint64_t v) { uint32_t k = 1; while (v >= 10) { v /= 10; k++; } return k; }
__device__ static inline uint32_t d_dec_len_signed (int64_t v) { return v < 0 ? 1 + d_dec_len ((uint64_t)(-v)) : d_dec_len ((uint64_t)v); }
__device__ static inline uint8_t *d_put_dec (uint8_t *o, uint64_t v) { const uint32_t k = d_dec_len (v); for (uint32_t i = k; i--; ) { o[i] = (uint8_t)('0' + v % 10); v /= 10; } return o + k; }
__device__ static inline uint8_t *d_put_dec_signed (uint8_t *o, int64_t v) { if (v < 0) { *o++ = '-'; v = -v; } return d_put_dec (o, (uint64_t)v); }

__device__ static inline uint32_t d_aux_elem_size (uint8_t t) { return (t == 'A' || t == 'c' || t == 'C') ? 1u : (t == 's' || t == 'S') ? 2u : (t == 'i' || t == 'I' || t == 'f') ? 4u : 0u; }
__device__ static inline int64_t d_aux_int (const uint8_t *p, uint8_t t)
{
    switch (t) {
        case 'c': return (int8_t)p[0];
        case 'C': return p[0];
        case 's': return (int16_t)d_le16 (p);
        case 'S': return d_le16 (p);
        case 'i': return (int32_t)d_le32 (p);
        default:  return d_le32 (p);       // 'I'
    }
}

// The optional fields of a record (bam_split_aux, src/bam_seg.c:187-224) as SAM text, each with its leading tab: TAG:A:c, TAG:i:n for
// every integer type, TAG:Z:string, TAG:H:hex, TAG:B:t,n,n.. Writes if out != NULL. Returns the length, or GZ_BAM_NONE for what is
// not handled here: a malformed field, or a float (f, B:f - the reference keeps those binary behind a special of their own, sam.h:863).
__device__ static inline uint32_t d_bam_aux_text (const uint8_t *aux, const uint8_t *after, uint8_t *out)
{
    uint32_t len = 0;
    uint8_t *o = out;
    while (aux < after) {
        if (after - aux < 4) return GZ_BAM_NONE;
        const uint8_t t = aux[2];
        if (o) { *o++ = '\t'; *o++ = aux[0]; *o++ = aux[1]; *o++ = ':'; }
        len += 4;
        if (t == 'Z' || t == 'H') {
            const uint8_t *s = aux + 3;
            uint32_t k = 0;
            while (s + k < after && s[k]) k++;
            if (s + k >= after) return GZ_BAM_NONE;
            if (o) { *o++ = t; *o++ = ':'; for (uint32_t i = 0; i < k; i++) *o++ = s[i]; }
            len += 2 + k;
            aux = s + k + 1;
        }
        else if (t == 'B') {
            if (after - aux < 8) return GZ_BAM_NONE;
            const uint8_t st = aux[3];
            const uint32_t w = d_aux_elem_size (st), cnt = d_le32 (aux + 4);
            if (!w || st == 'f' || st == 'A' || (uint64_t)cnt * w > (uint64_t)(after - aux - 8)) return GZ_BAM_NONE;
            if (o) { *o++ = 'B'; *o++ = ':'; *o++ = st; }
            len += 3;
            for (uint32_t i = 0; i < cnt; i++) {
                const int64_t v = d_aux_int (aux + 8 + (size_t)i * w, st);
                if (o) { *o++ = ','; o = d_put_dec_signed (o, v); }
                len += 1 + d_dec_len_signed (v);
            }
            aux += 8 + (size_t)cnt * w;
        }
        else if (t == 'A') {
            if (o) { *o++ = 'A'; *o++ = ':'; *o++ = aux[3]; }
            len += 3;
            aux += 4;
        }
        else {
            const uint32_t w = d_aux_elem_size (t);
            if (!w || t == 'f' || (uint32_t)(after - aux) < 3 + w) return GZ_BAM_NONE;
            const int64_t v = d_aux_int (aux + 3, t);
            if (o) { *o++ = 'i'; *o++ = ':'; o = d_put_dec_signed (o, v); }
            len += 2 + d_dec_len_signed (v);
            aux += 3 + w;
        }
    }
    return len;
}

struct GzdBamRec {
    const uint8_t *a, *after, *name, *cigar, *seq, *qual, *aux;
    int32_t ref_id, pos, next_ref, next_pos, tlen;
    uint32_t l_read_name, mapq, n_cigar, flag, l_seq;
    bool ok;
};

__device__ static inline GzdBamRec d_bam_rec (const GzdBamText &T, uint32_t r)
{
    GzdBamRec R;
    const uint8_t *a = T.bam + T.rec_off[r];
    const uint32_t bs = d_le32 (a);
    R.a = a; R.after = a + 4 + bs;
    R.ref_id = (int32_t)d_le32 (a + 4); R.pos = (int32_t)d_le32 (a + 8); R.l_read_name = a[12]; R.mapq = a[13];
    R.n_cigar = d_le16 (a + 16); R.flag = d_le16 (a + 18); R.l_seq = d_le32 (a + 20);
    R.next_ref = (int32_t)d_le32 (a + 24); R.next_pos = (int32_t)d_le32 (a + 28); R.tlen = (int32_t)d_le32 (a + 32);
    R.name = a + 36; R.cigar = R.name + R.l_read_name; R.seq = R.cigar + 4 * (size_t)R.n_cigar; R.qual = R.seq + (R.l_seq + 1) / 2; R.aux = R.qual + R.l_seq;
    // (bam_seg.c:444-447,459-460: the fields must fit the record, the reference ids the header)
    R.ok = R.l_read_name >= 1 && R.l_seq <= bs && (uint64_t)32 + R.l_read_name + 4ull * R.n_cigar + (R.l_seq + 1) / 2 + R.l_seq <= bs &&
           R.ref_id >= -1 && R.ref_id < T.n_ref && R.next_ref >= -1 && R.next_ref < T.n_ref;
    return R;
}

__device__ static inline uint32_t d_ref_name_len (const GzdBamText &T, int32_t id) { return T.ref_name_off[id + 1] - T.ref_name_off[id]; }

// the length of the record's line with its newline; GZ_BAM_NONE if it cannot be written
__device__ static inline uint32_t d_bam_line_len (const GzdBamText &T, const GzdBamRec &R)
{
    if (!R.ok) return GZ_BAM_NONE;
    uint64_t len = (R.l_read_name - 1) + 1;
    len += d_dec_len (R.flag) + 1;
    len += (R.ref_id < 0 ? 1 : d_ref_name_len (T, R.ref_id)) + 1;
    len += d_dec_len_signed ((int64_t)R.pos + 1) + 1;
    len += d_dec_len (R.mapq) + 1;
    if (!R.n_cigar) len += 1;
    else for (uint32_t i = 0; i < R.n_cigar; i++) len += d_dec_len (d_le32 (R.cigar + 4 * (size_t)i) >> 4) + 1;
    len += 1;
    len += (R.next_ref < 0 ? 1 : (R.next_ref == R.ref_id ? 1 : d_ref_name_len (T, R.next_ref))) + 1;
    len += d_dec_len_signed ((int64_t)R.next_pos + 1) + 1;
    len += d_dec_len_signed (R.tlen) + 1;
    len += (R.l_seq ? R.l_seq : 1) + 1;
    len += (R.l_seq && R.qual[0] != 0xff) ? R.l_seq : 1;
    const uint32_t al = d_bam_aux_text (R.aux, R.after, NULL);
    if (al == GZ_BAM_NONE) return GZ_BAM_NONE;
    len += al + 1;
    return len > 0xfffffff0ull ? GZ_BAM_NONE : (uint32_t)len;
}

// grid (tiles of 256 records)
__global__ void __launch_bounds__(256) k_bam_len (GzdBamText T)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    uint32_t len = 0;
    if (r < T.n_rec) {
        const GzdBamRec R = d_bam_rec (T, r);
        len = d_bam_line_len (T, R);
        if (len == GZ_BAM_NONE) { atomicMin (&T.result->first_bad, r); len = 0; }
        T.len[r] = len;
    }
    uint64_t total;
    (void)d_wg_scan_u64 (len, threadIdx.x, &total);
    if (!threadIdx.x) T.tile[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_bam_scan (GzdBamText T)
{
    const uint64_t total = d_wg_scan_array (T.tile, (T.n_rec + 255) / 256, threadIdx.x);
    if (!threadIdx.x) {
        T.result->text_len = total; T.result->n_records = T.n_rec; T.result->n_rewalked = 0; T.result->reserved = 0;
        T.result->status = T.result->first_bad != GZ_BAM_NONE ? GZ_ST_CORRUPT : total > T.text_cap ? GZ_ST_TOO_SMALL : GZ_ST_OK;
        if (T.line_off) T.line_off[T.n_rec] = (uint32_t)total;
    }
}

static __device__ const char d_bam_bases[17] = "=ACMGRSVTWYHKDBN";        // bam_base_codes (src/bam_seq.c:15)
static __device__ const char d_bam_cigar_ops[17] = "MIDNSHP=Xabcdefg";    // cigar_op_to_char (src/sam_cigar.c:23)

// grid (tiles of 256 records): the thread of a record writes everything but SEQ / QUAL, then the wave writes those of its 64
// records one after the other (4 bases or scores per lane and round)
__global__ void __launch_bounds__(256) k_bam_write (GzdBamText T)
{
    if (T.result->status != GZ_ST_OK) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t r = blockIdx.x * 256 + tid;
    const bool on = r < T.n_rec;
    const uint32_t len = on ? T.len[r] : 0;
    uint64_t total;
    const uint64_t at = T.tile[blockIdx.x] + d_wg_scan_u64 (len, tid, &total);
    uint32_t seq_at = 0, qual_at = 0, l_seq = 0, roff = 0;
    bool has_qual = false;
    if (on) {
        if (T.line_off) T.line_off[r] = (uint32_t)at;
        const GzdBamRec R = d_bam_rec (T, r);
        uint8_t *o = T.text + at;
        for (uint32_t i = 0; i + 1 < R.l_read_name; i++) *o++ = R.name[i];
        *o++ = '\t'; o = d_put_dec (o, R.flag); *o++ = '\t';
        if (R.ref_id < 0) *o++ = '*';
        else { const uint8_t *nm = T.ref_names + T.ref_name_off[R.ref_id]; const uint32_t k = d_ref_name_len (T, R.ref_id); for (uint32_t i = 0; i < k; i++) *o++ = nm[i]; }
        *o++ = '\t'; o = d_put_dec_signed (o, (int64_t)R.pos + 1); *o++ = '\t'; o = d_put_dec (o, R.mapq); *o++ = '\t';
        if (!R.n_cigar) *o++ = '*';
        else for (uint32_t i = 0; i < R.n_cigar; i++) { const uint32_t op = d_le32 (R.cigar + 4 * (size_t)i); o = d_put_dec (o, op >> 4); *o++ = (uint8_t)d_bam_cigar_ops[op & 15]; }
        *o++ = '\t';
        if (R.next_ref < 0) *o++ = '*';
        else if (R.next_ref == R.ref_id) *o++ = '=';
        else { const uint8_t *nm = T.ref_names + T.ref_name_off[R.next_ref]; const uint32_t k = d_ref_name_len (T, R.next_ref); for (uint32_t i = 0; i < k; i++) *o++ = nm[i]; }
        *o++ = '\t'; o = d_put_dec_signed (o, (int64_t)R.next_pos + 1); *o++ = '\t'; o = d_put_dec_signed (o, R.tlen); *o++ = '\t';
        l_seq = R.l_seq; has_qual = R.l_seq && R.qual[0] != 0xff; roff = T.rec_off[r];
        seq_at = (uint32_t)(o - (T.text + at));
        if (!l_seq) *o = '*';
        o += l_seq ? l_seq : 1;
        *o++ = '\t';
        qual_at = (uint32_t)(o - (T.text + at));
        if (!has_qual) *o = '*';
        o += has_qual ? l_seq : 1;
        o += d_bam_aux_text (R.aux, R.after, o);
        *o = '\n';
    }
    // SEQ and QUAL, a record at a time by the whole wave
    uint64_t m = __ballot (on && l_seq);
    while (m) {
        const int src = __ffsll ((unsigned long long)m) - 1;
        m &= m - 1;
        const uint32_t n = (uint32_t)__shfl ((int)l_seq, src), ro = (uint32_t)__shfl ((int)roff, src);
        const uint32_t sa = (uint32_t)__shfl ((int)seq_at, src), qa = (uint32_t)__shfl ((int)qual_at, src);
        const bool hq = __shfl ((int)has_qual, src) != 0;
        const uint64_t lat = ((uint64_t)(uint32_t)__shfl ((int)(uint32_t)(at >> 32), src) << 32) | (uint32_t)__shfl ((int)(uint32_t)at, src);
        const uint8_t *a = T.bam + ro;
        const uint8_t *seq = a + 36 + a[12] + 4 * (size_t)d_le16 (a + 16), *qual = seq + (n + 1) / 2;
        uint8_t *os = T.text + lat + sa, *oq = T.text + lat + qa;
        for (uint32_t b = (uint32_t)lane * 4; b < n; b += 256) {                    // bases b .. b + 3 <- bytes b / 2, b / 2 + 1
            const uint8_t s0 = seq[b >> 1], s1 = b + 2 < n ? seq[(b >> 1) + 1] : 0;
            os[b] = (uint8_t)d_bam_bases[s0 >> 4];
            if (b + 1 < n) os[b + 1] = (uint8_t)d_bam_bases[s0 & 15];
            if (b + 2 < n) os[b + 2] = (uint8_t)d_bam_bases[s1 >> 4];
            if (b + 3 < n) os[b + 3] = (uint8_t)d_bam_bases[s1 & 15];
            if (hq) for (uint32_t k = 0; k < 4 && b + k < n; k++) oq[b + k] = (uint8_t)(qual[b + k] + 33);   // bam_rewrite_qual
        }
    }
}

/* tests/c/zip_fastq.c -- a plain C11 host program against include/genozip_amd.h (no Python, no HIP headers): the host
 * orchestration of the north star "stays in C and calls HIP through a thin C-ABI". It builds a small paired FASTQ text,
 * hands it to the VBlock compute driver (gz_fastq_zip_vblocks: text -> seg columns -> dictionary merge -> b250 / local
 * generation -> codecs -> sections), walks the resulting z_data, decodes every section again on the device
 * (gz_vb_uncompress checks each adler32) and compares the decoded QUAL local with the text's quality lines.
 *
 *   gcc -std=c11 -Wall -Wextra -pedantic -I include tests/c/zip_fastq.c -o zip_fastq -L genozip_amd -lgenozip_amd -Wl,-rpath,$PWD/genozip_amd
 *   ./zip_fastq [n_reads]          exit code 0 = everything checked out; 77 = no GPU (gz_create refused: there is no CPU fallback)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "genozip_amd.h"

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd (void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }

static void did (uint8_t id[8], const char *tag, int dtype)
{
    memset (id, 0, 8);
    memcpy (id, tag, strlen (tag) > 8 ? 8 : strlen (tag));
    id[0] = dtype == 0 ? (id[0] & 0x3f) : (id[0] | 0xc0);             /* dict_id_make, src/dict_id.c:34-36 */
}

static uint32_t slen (const uint8_t *s) { return s ? (uint32_t)strlen ((const char *)s) : 0; }

#define FAIL(...) do { fprintf (stderr, __VA_ARGS__); fprintf (stderr, "\n"); return 1; } while (0)

int main (int argc, char **argv)
{
    const uint32_t n_reads = argc > 1 ? (uint32_t)atoi (argv[1]) : 3000, L = 150;
    int err = 0;
    GzHandle *h = gz_create (0, NULL, &err);
    if (!h) { fprintf (stderr, "gz_create: %d (no GPU: this library has no CPU fallback)\n", err); return 77; }
    printf ("%s\n", gz_version ());

    /* the plan: Illumina-7 QNAME + Illum-2bc QNAME2 (src/qname_flavors.h:40-49,1095,1205), see genozip_amd/fastq.py */
    static const uint8_t con1[] = "\4<illumina-7 container>", con2[] = "\4<illum-2bc container>", sqb[] = "\10<unaligned SEQ>", top[] = "\4<fastq toplevel>",
                         eol[] = "\n", l3[] = "\10<line3 = empty>", delta[] = "\5$";
    GzFastqCtx C[17];
    memset (C, 0, sizeof (C));
    int n = 0;
#define CTX(tag, dt, did_, kind_, item_, flags_, snip_, pi) do { did (C[n].dict_id, tag, dt); C[n].did_i = did_; C[n].kind = kind_; C[n].item = item_; \
    C[n].flags = flags_; C[n].snip = (const uint8_t *)(snip_); C[n].snip_len = slen (C[n].snip); C[n].pair_identical = pi; \
    C[n].no_stons = (tag[0] == 'Q' || tag[0] == 'q') && strcmp (tag, "QUAL"); n++; } while (0)
    CTX ("QNAME", 0, 1, GZ_FQ_CONST, 0, 0, con1, 1);
    CTX ("Q0NAME", 1, 2, GZ_FQ_ITEM_TEXT, 0, 0, NULL, 1);
    CTX ("Q1NAME", 1, 3, GZ_FQ_ITEM_INT, 1, 0, NULL, 1);
    CTX ("Q2NAME", 1, 4, GZ_FQ_ITEM_TEXT, 2, 0, NULL, 1);
    CTX ("Q3NAME", 1, 5, GZ_FQ_ITEM_DELTA, 3, 1, delta, 1);
    CTX ("Q4NAME", 1, 6, GZ_FQ_ITEM_DELTA, 4, 1, delta, 1);
    CTX ("QNAME2", 0, 18, GZ_FQ_CONST, 0, 0, con2, 1);
    CTX ("q0NAME", 1, 19, GZ_FQ_ITEM_TEXT, 5, 0, NULL, 1);
    CTX ("q1NAME", 1, 20, GZ_FQ_ITEM_TEXT, 6, 0, NULL, 1);
    CTX ("q2NAME", 1, 21, GZ_FQ_ITEM_TEXT, 7, 0, NULL, 1);
    CTX ("SQBITMAP", 0, 40, GZ_FQ_CONST, 0, 0, sqb, 0); C[n - 1].pair_assisted_b250 = 1;
    CTX ("NONREF_X", 0, 42, GZ_FQ_SEQ, 0, 0, NULL, 0); C[n - 1].local_dep = 1;
    CTX ("QUAL", 0, 80, GZ_FQ_QUAL, 0, 0, NULL, 0);
    CTX ("TOPLEVEL", 0, 90, GZ_FQ_CONST, 0, 0, top, 1);
    CTX ("E1L", 0, 96, GZ_FQ_CONST, 0, 0, eol, 1);
    CTX ("E2L", 0, 97, GZ_FQ_CONST, 0, 0, eol, 1);
    CTX ("LINE3", 0, 98, GZ_FQ_CONST, 0, 0, l3, 1);
    GzFastqPlan P;
    memset (&P, 0, sizeof (P));
    P.ctxs = C; P.n_ctxs = (uint32_t)n; memcpy (P.seps, ":::: :+", 7);
    { const uint8_t cnt[7] = { 3, 1, 1, 1, 1, 3, 1 }; memcpy (P.sep_counts, cnt, 7); }
    P.n_seps = 7; P.paired = 1;
    GzZipFile *f = gz_zip_open (h, &P);
    if (!f) FAIL ("gz_zip_open failed");

    /* the text: R1 then R2, one VBlock each; R2 names its R1 VBlock */
    const size_t cap = (size_t)2 * n_reads * (64 + 2 * (L + 1) + 8) + 64;
    char *text = malloc (cap), *quals = malloc ((size_t)2 * n_reads * L + 1);
    if (!text || !quals) FAIL ("out of memory");
    size_t at = 0, qat = 0, mate_at[3] = { 0, 0, 0 };
    for (int mate = 1; mate <= 2; mate++) {
        rng_state = 88172645463325252ull;                               /* same names in both mates */
        uint64_t y = 1000;
        for (uint32_t i = 0; i < n_reads; i++) {
            const uint32_t tile = 1101 + i * 40 / n_reads, x = 1000 + rnd () % 30000;
            y += rnd () % 9;
            at += (size_t)sprintf (text + at, "@A00123:45:HXXXXXXXX:%u:%u:%u:%llu %d:N:0:ACGTACGT+TGCATGCA\n", 1 + i * 4 / n_reads, tile, x, (unsigned long long)y, mate);
            for (uint32_t k = 0; k < L; k++) text[at++] = "ACGT"[(rnd () >> (2 * mate)) & 3];
            text[at++] = '\n'; text[at++] = '+'; text[at++] = '\n';
            uint32_t q = 38;
            for (uint32_t k = 0; k < L; k++) { q = q + rnd () % 3 - 1; q = q < 2 ? 2 : q > 41 ? 41 : q; text[at] = quals[qat++] = (char)(33 + (q + mate) % 42); at++; }
            text[at++] = '\n';
        }
        mate_at[mate] = at;
    }
    uint8_t *d_text = gz_dev_alloc (h, at + 64);
    if (!d_text || gz_upload (h, d_text, text, at) != GZ_OK) FAIL ("upload: %s", gz_last_error (h));
    GzFastqVB vb[2];
    memset (vb, 0, sizeof (vb));
    vb[0].text_off = 0;          vb[0].text_len = mate_at[1];              vb[0].vblock_i = 1; vb[0].r1 = -1;
    vb[1].text_off = mate_at[1]; vb[1].text_len = mate_at[2] - mate_at[1]; vb[1].vblock_i = 2; vb[1].r1 = 0;
    if (gz_fastq_zip_vblocks (f, d_text, at, vb, 2) != GZ_OK) FAIL ("gz_fastq_zip_vblocks: %s", gz_last_error (h));

    for (int v = 0; v < 2; v++) {
        if (vb[v].n_reads != n_reads || vb[v].n_bases != (uint64_t)n_reads * L || vb[v].seq_has_x) FAIL ("VBlock %d: reads / bases", v + 1);
        uint8_t *z = malloc (vb[v].z_len);
        if (gz_download (h, z, vb[v].z_data, vb[v].z_len) != GZ_OK) FAIL ("download");
        /* walk the sections: SectionHeaderCtx is 40 bytes, big endian (src/sections.h:146-167,419-435) */
        uint64_t total = 0, p = 84;
        int n_sec = 0, qual_i = -1;
        uint8_t qid[8]; did (qid, "QUAL", 0);
        while (p < vb[v].z_len) {
            const uint32_t clen = (uint32_t)z[p + 12] << 24 | z[p + 13] << 16 | z[p + 14] << 8 | z[p + 15], ulen = (uint32_t)z[p + 16] << 24 | z[p + 17] << 16 | z[p + 18] << 8 | z[p + 19];
            if (!memcmp (z + p + 32, qid, 8)) qual_i = n_sec;
            total += ulen; p += 40 + clen; n_sec++;
        }
        if (p != vb[v].z_len || qual_i < 0) FAIL ("VBlock %d: section walk", v + 1);
        uint8_t *d_out = gz_dev_alloc (h, total + 64), *out = malloc (total + 1);
        uint64_t offs[64]; uint32_t ns = 0;
        if (gz_vb_uncompress (h, vb[v].z_data, vb[v].z_len, d_out, total, offs, 63, &ns) != GZ_OK) FAIL ("gz_vb_uncompress: %s", gz_last_error (h));
        if ((int)ns != n_sec || gz_download (h, out, d_out, total) != GZ_OK) FAIL ("decode");
        if (offs[qual_i + 1] - offs[qual_i] != (uint64_t)n_reads * L || memcmp (out + offs[qual_i], quals + (size_t)v * n_reads * L, (size_t)n_reads * L))
            FAIL ("VBlock %d: QUAL does not come back", v + 1);
        printf ("VBlock %d: %u reads, %llu -> %llu bytes, %d sections, QUAL round trip ok\n", v + 1, vb[v].n_reads, (unsigned long long)vb[v].text_len, (unsigned long long)vb[v].z_len, n_sec);
        if (v == 1 && n_sec >= 6) FAIL ("R2 kept sections that are identical to R1's");
        gz_dev_free (h, d_out); free (out); free (z);
    }
    GzZctxView zv;
    if (gz_zctx_view (gz_zip_zctx (f, 3), &zv) != GZ_OK || zv.n_words != 40) FAIL ("tile dictionary: %u words", zv.n_words);
    gz_dev_free (h, d_text);
    gz_zip_close (f);
    gz_destroy (h);
    free (text); free (quals);
    printf ("OK\n");
    return 0;
}

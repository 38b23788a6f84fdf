/* ref_shim.c -- TEST INFRASTRUCTURE ONLY (not shipped, not linked into the product library).
 *
 * Five-symbol shim that lets the reference's vendored htscodecs sources
 * (/root/reference/src/htscodecs/{rANS_static4x16pr,arith_dynamic,pack,rle}.c, BSD licensed, compiled
 * IN PLACE - never copied into this repo) link into oracle/_ref/libhtsref.so without the rest of Genozip.
 *
 * The reference routes its scratch allocations through codec_alloc()/codec_free() (src/codec.h:143-147,
 * src/codec.c:30-63) and MALLOC/FREE (src/buf_struct.h:222-227) and its assertions through
 * error_assert_failed() (src/genozip.h:747). Here they are plain malloc/free/abort.
 *
 * This file is authored for this repo; it contains no reference code.
 */
#include <stdint.h>
#include <stdbool.h>
#include <stdlib.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

void *codec_alloc_do (void *vb, uint64_t size, float grow_at_least_factor, unsigned *buf_i,
                      const char *func, uint32_t code_line)
{
    (void)vb; (void)grow_at_least_factor; (void)func; (void)code_line;
    if (buf_i) *buf_i = 0;
    return malloc (size ? size : 1);
}

void codec_free_do (void *vb, void *addr, const char *func, uint32_t code_line)
{
    (void)vb; (void)func; (void)code_line;
    free (addr);
}

void *buf_low_level_malloc (size_t size, bool zero, const char *func, uint32_t code_line)
{
    (void)func; (void)code_line;
    return zero ? calloc (size ? size : 1, 1) : malloc (size ? size : 1);
}

void buf_low_level_free (void *p, const char *func, uint32_t code_line)
{
    (void)func; (void)code_line;
    free (p);
}

void error_assert_failed (const char *func, uint32_t code_line, const char *format, ...)
{
    va_list ap;
    va_start (ap, format);
    fprintf (stderr, "htsref assertion failed in %s:%u: ", func, code_line);
    vfprintf (stderr, format, ap);
    fprintf (stderr, "\n");
    va_end (ap);
    abort ();
}

/* ---- flat entry points used by the python test harness (ctypes) and by bench.py's cpu_baseline ---- */

extern unsigned char *rans_compress_to_4x16 (void *vb, unsigned char *in, unsigned int in_size,
                                             unsigned char *out, unsigned int *out_size, int order);
extern unsigned char *rans_uncompress_to_4x16 (void *vb, unsigned char *in, unsigned int in_size,
                                               unsigned char *out, unsigned int *out_size);
extern unsigned int rans_compress_bound_4x16 (unsigned int size, int order);
extern unsigned char *arith_compress_to (void *vb, unsigned char *in, unsigned int in_size,
                                         unsigned char *out, unsigned int *out_size, int order);
extern unsigned char *arith_uncompress_to (void *vb, unsigned char *in, unsigned int in_size,
                                           unsigned char *out, unsigned int *out_size);
extern unsigned int arith_compress_bound (unsigned int size, int order);

/* returns compressed length, or -1 if the reference signalled "output buffer too small" */
long htsref_rans_compress (const unsigned char *in, unsigned in_size, unsigned char *out, unsigned out_cap, int order)
{
    unsigned out_size = out_cap;
    return rans_compress_to_4x16 (NULL, (unsigned char *)in, in_size, out, &out_size, order) ? (long)out_size : -1;
}

long htsref_arith_compress (const unsigned char *in, unsigned in_size, unsigned char *out, unsigned out_cap, int order)
{
    unsigned out_size = out_cap;
    return arith_compress_to (NULL, (unsigned char *)in, in_size, out, &out_size, order) ? (long)out_size : -1;
}

long htsref_rans_uncompress (const unsigned char *in, unsigned in_size, unsigned char *out, unsigned out_len)
{
    unsigned out_size = out_len;
    return rans_uncompress_to_4x16 (NULL, (unsigned char *)in, in_size, out, &out_size) ? (long)out_size : -1;
}

long htsref_arith_uncompress (const unsigned char *in, unsigned in_size, unsigned char *out, unsigned out_len)
{
    unsigned out_size = out_len;
    return arith_uncompress_to (NULL, (unsigned char *)in, in_size, out, &out_size) ? (long)out_size : -1;
}

unsigned htsref_rans_bound  (unsigned size, int order) { return rans_compress_bound_4x16 (size, order); }
unsigned htsref_arith_bound (unsigned size, int order) { return arith_compress_bound (size, order); }

/* ---- many independent streams on a pthread pool: the multithreaded CPU baseline of bench.py ("kind": "reference").
 * Mirrors what Genozip's dispatcher does for this path: one compute thread per VBlock-sized unit of work
 * (src/dispatcher.c:544-618), here one codec call per task. codecs use Genozip's order bytes
 * (src/codec_htscodecs.c:17-20). Every task is a codec call: the caller leaves out the sections comp_compress stores raw (simple
 * codecs under 50 bytes, src/compressor.c:56-58) - the sub-codec stream of a complex codec (CODEC_DOMQ's QUAL) is coded whatever
 * its length (codec_domq.c:508-520 calls the sub-codec directly). ---- */
#include <pthread.h>

typedef struct {
    int n; const int *is_arith; const int *orders; const unsigned char *const *ins; const unsigned *in_lens;
    unsigned char *const *outs; const unsigned *out_caps; long *out_lens; int next; pthread_mutex_t mu; int replicas;
} HtsRefMany;

static void *htsref_many_worker (void *arg)
{
    HtsRefMany *j = arg;
    unsigned char *scratch = NULL; unsigned scratch_cap = 0;       /* (replicas beyond the first write here: same work, the output is not kept) */
    for (;;) {
        pthread_mutex_lock (&j->mu);
        int t = j->next++;
        pthread_mutex_unlock (&j->mu);
        if (t >= j->n * j->replicas) { free (scratch); return NULL; }
        const int i = t % j->n;
        unsigned char *out = j->outs[i];
        if (t >= j->n) {
            if (scratch_cap < j->out_caps[i]) { free (scratch); scratch = malloc (j->out_caps[i]); scratch_cap = scratch ? j->out_caps[i] : 0; }
            if (!scratch) { j->out_lens[i] = -1; continue; }
            out = scratch;
        }
        const long l = j->is_arith[i] ? htsref_arith_compress (j->ins[i], j->in_lens[i], out, j->out_caps[i], j->orders[i])
                                      : htsref_rans_compress  (j->ins[i], j->in_lens[i], out, j->out_caps[i], j->orders[i]);
        if (t < j->n || l < 0) j->out_lens[i] = l;
    }
}

int htsref_compress_many_rep (int n, const int *is_arith, const int *orders, const unsigned char *const *ins, const unsigned *in_lens,
                              unsigned char *const *outs, const unsigned *out_caps, long *out_lens, int n_threads, int replicas);
int htsref_compress_many (int n, const int *is_arith, const int *orders, const unsigned char *const *ins, const unsigned *in_lens,
                          unsigned char *const *outs, const unsigned *out_caps, long *out_lens, int n_threads)
{
    return htsref_compress_many_rep (n, is_arith, orders, ins, in_lens, outs, out_caps, out_lens, n_threads, 1);
}

/* every task `replicas` times (the task list replica after replica): a file with more VBlocks of the same kind - tasks >= 4 x threads keep
 * every thread busy until the end, which 150 sections on 256 threads do not */
int htsref_compress_many_rep (int n, const int *is_arith, const int *orders, const unsigned char *const *ins, const unsigned *in_lens,
                              unsigned char *const *outs, const unsigned *out_caps, long *out_lens, int n_threads, int replicas)
{
    HtsRefMany j = { n, is_arith, orders, ins, in_lens, outs, out_caps, out_lens, 0, PTHREAD_MUTEX_INITIALIZER, replicas < 1 ? 1 : replicas };
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    pthread_t th[1024];
    for (int t = 1; t < n_threads; t++) pthread_create (&th[t], NULL, htsref_many_worker, &j);
    htsref_many_worker (&j);
    for (int t = 1; t < n_threads; t++) pthread_join (th[t], NULL);
    for (int i = 0; i < n; i++) if (out_lens[i] < 0) return -1;
    return 0;
}

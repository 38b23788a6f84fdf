/* oracle/ref_lzma_shim.c -- TEST INFRASTRUCTURE ONLY (see oracle/Makefile, target "ref").
 *
 * The reference vendors the 7-zip LZMA SDK (public domain) under src/lzma/; its CODEC_LZMA (src/codec_lzma.c:95-165) is the
 * sub-codec of CODEC_ACGT, i.e. what NONREF - the 2-bit packed SEQ - is finally compressed with. LZMA is sequential host work
 * and outside this repo's path: the product hands the packed bytes back to the caller. For end-to-end tests
 * against the reference's genounzip (tests/test_e2e_genounzip.py) the NONREF section still has to exist, and made by the
 * reference's own encoder it is byte for byte what the reference would write. This file calls the SDK, compiled where it lies,
 * with the parameters codec_lzma_compress sets: level 5, fb 273, an end mark, dictSize = MIN (length, segconf.vb_size); 5 bytes of
 * properties in front of the stream. Allocation goes to malloc (the SDK was patched to use codec_alloc). Nothing here is product code.
 */
#include <stdint.h>
#include <stdbool.h>
#include <stdlib.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include "genozip.h"
#include "lzma/LzmaEnc.h"
#include "lzma/LzmaDec.h"

void *codec_alloc_do (VBlockP vb, uint64_t size, float grow_at_least_factor, unsigned *buf_i, FUNCLINE)
{
    (void)vb; (void)grow_at_least_factor; (void)func; (void)code_line;
    if (buf_i) *buf_i = 0;
    return malloc (size ? size : 1);
}
void codec_free_do (VBlockP vb, void *addr, FUNCLINE) { (void)vb; (void)func; (void)code_line; free (addr); }
void *buf_low_level_malloc (size_t size, bool zero, FUNCLINE) { (void)func; (void)code_line; return zero ? calloc (size ? size : 1, 1) : malloc (size ? size : 1); }
void buf_low_level_free (void *p, FUNCLINE) { (void)func; (void)code_line; free (p); }
void error_assert_failed (FUNCLINE, rom format, ...)
{
    va_list ap; va_start (ap, format);
    fprintf (stderr, "lzmaref assertion failed in %s:%u: ", func, code_line); vfprintf (stderr, format, ap); fprintf (stderr, "\n");
    va_end (ap); abort ();
}

/* -> compressed length (5 property bytes + stream), -1: out_cap too small, -2: another error */
long lzmaref_compress (const uint8_t *in, uint32_t in_len, uint64_t vb_size, uint8_t *out, uint32_t out_cap)
{
    CLzmaEncProps props;
    LzmaEncProps_Init (&props);
    props.level = 5; props.fb = 273; props.writeEndMark = true;
    props.dictSize = in_len < vb_size ? in_len : (uint32_t)vb_size;
    char *handle = malloc (LzmaEnc_LzmaHandleSize ());
    if (!handle || out_cap < LZMA_PROPS_SIZE) { free (handle); return -1; }
    LzmaEnc_Create (handle, NULL, NULL);
    long ret = -2;
    SizeT props_size = LZMA_PROPS_SIZE;
    if (LzmaEnc_SetProps (handle, &props) == SZ_OK && LzmaEnc_WriteProperties (handle, out, &props_size) == SZ_OK && props_size == LZMA_PROPS_SIZE) {
        SizeT clen = (SizeT)out_cap - LZMA_PROPS_SIZE;
        const SRes res = LzmaEnc_MemEncode (handle, out + LZMA_PROPS_SIZE, &clen, in, in_len, true);
        ret = res == SZ_OK ? (long)clen + LZMA_PROPS_SIZE : res == SZ_ERROR_OUTPUT_EOF ? -1 : -2;
    }
    LzmaEnc_Destroy (handle);
    free (handle);
    return ret;
}

/* -> 0 if `in` decodes to exactly out_len bytes ending with the end mark */
int lzmaref_uncompress (const uint8_t *in, uint32_t in_len, uint8_t *out, uint64_t out_len)
{
    if (in_len < LZMA_PROPS_SIZE) return -1;
    ELzmaStatus status;
    SizeT clen = (SizeT)in_len - LZMA_PROPS_SIZE, ulen = out_len;
    const SRes r = LzmaDecode (NULL, out, &ulen, in + LZMA_PROPS_SIZE, &clen, in, LZMA_PROPS_SIZE, LZMA_FINISH_END, &status);
    return r == SZ_OK && status == LZMA_STATUS_FINISHED_WITH_MARK && ulen == out_len ? 0 : -1;
}

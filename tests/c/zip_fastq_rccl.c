/* tests/c/zip_fastq_rccl.c -- the N-GPU route of the path for a host written in C (BASELINE north_star: "host orchestration stays in C";
 * VBlocks shard over the GPUs of a node, RCCL over xGMI only for the exchange). One process per GPU. What travels is defined by
 * the C-ABI of include/genozip_amd.h - opaque byte strings:
 *     the merge blob of gz_fastq_zip_seg      (the new words of this rank's VBlocks: a few KB)      -> every rank   (all-gather)
 *     the votes      of gz_fastq_zip_merge    (codec choices for contexts the file has none for)     -> every rank   (all-gather)
 *     the z_data     of gz_fastq_zip_collect  (the finished VBlocks, one device buffer + offsets)    -> the writer   (send / recv)
 * and the transport is RCCL's own C API (rccl.h: ncclAllGather, ncclSend / ncclRecv in a group) on the device buffers themselves - the
 * counterpart of genozip_amd/shard.py, which does the same through torch.distributed for the tests and bench.py. The reference's analogue is
 * the dispatcher handing finished VBlocks to the writer in order (src/dispatcher.c:544-618, src/zip.c:765).
 *
 * tests/test_abi.py compiles this file against include/genozip_amd.h and <rccl/rccl.h> (syntax and types: the header is all a C host
 * needs); it is run where a node has several GPUs:   mpirun -n 8 ./zip_fastq_rccl   (rank r takes GPU r; the ncclUniqueId goes round by
 * whatever the host has - here a file).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "genozip_amd.h"

#define CK(x)   do { if ((x) != hipSuccess)  { fprintf (stderr, "%s failed\n", #x); return -1; } } while (0)
#define NC(x)   do { if ((x) != ncclSuccess) { fprintf (stderr, "%s failed\n", #x); return -1; } } while (0)
#define GZ(x)   do { if ((x) != GZ_OK)       { fprintf (stderr, "%s failed\n", #x); return -1; } } while (0)

/* every rank's byte string to every rank: lengths first, then the strings padded to the longest (they are small) */
static int all_gather_bytes (ncclComm_t comm, hipStream_t st, int world, const void *mine, uint64_t mine_len, uint8_t **all_host, uint64_t *lens_host)
{
    uint64_t *d_lens; CK (hipMalloc ((void **)&d_lens, (size_t)(world + 1) * 8));
    CK (hipMemcpyAsync (d_lens + world, &mine_len, 8, hipMemcpyHostToDevice, st));
    NC (ncclAllGather (d_lens + world, d_lens, 1, ncclUint64, comm, st));
    CK (hipMemcpyAsync (lens_host, d_lens, (size_t)world * 8, hipMemcpyDeviceToHost, st));
    CK (hipStreamSynchronize (st));
    uint64_t cap = 1; for (int r = 0; r < world; r++) if (lens_host[r] > cap) cap = lens_host[r];
    uint8_t *d_buf; CK (hipMalloc ((void **)&d_buf, (size_t)(world + 1) * cap));
    CK (hipMemcpyAsync (d_buf + (size_t)world * cap, mine, mine_len, hipMemcpyHostToDevice, st));
    NC (ncclAllGather (d_buf + (size_t)world * cap, d_buf, cap, ncclUint8, comm, st));
    *all_host = malloc ((size_t)world * cap);
    CK (hipMemcpyAsync (*all_host, d_buf, (size_t)world * cap, hipMemcpyDeviceToHost, st));
    CK (hipStreamSynchronize (st));
    CK (hipFree (d_lens)); CK (hipFree (d_buf));
    return (int)cap;                                               /* rank r's string: *all_host + r * cap, lens_host[r] bytes */
}

/* one call of the VBlock compute driver on a file whose VBlocks are dealt out over the ranks: vbs = THIS rank's VBlocks (vblock_i as in the file) */
int zip_call_on_rank (GzZipFile *zf, ncclComm_t comm, hipStream_t st, int rank, int world, int writer,
                      uint8_t *dev_text, uint64_t text_len, GzFastqVB *vbs, int n_vbs, uint8_t *dev_z, uint64_t z_cap, uint64_t *z_offsets /* n_vbs + 1 */,
                      uint8_t *dev_gathered /* writer: room for every rank's z_data */, uint64_t *gathered_len /* writer: [world] */)
{
    /* seg -> everybody's merge blobs -> the ordered merge replayed on every rank -> everybody's votes -> finish */
    const void *blob, *votes; uint64_t blob_len, votes_len;
    GZ (gz_fastq_zip_seg (zf, dev_text, text_len, vbs, n_vbs, &blob, &blob_len));
    uint8_t *all; uint64_t lens[64]; const void *ptr[64];
    int cap = all_gather_bytes (comm, st, world, blob, blob_len, &all, lens);
    if (cap < 0) return -1;
    for (int r = 0; r < world; r++) ptr[r] = all + (size_t)r * (size_t)cap;
    GZ (gz_fastq_zip_merge (zf, ptr, lens, world, &votes, &votes_len));
    free (all);
    cap = all_gather_bytes (comm, st, world, votes, votes_len, &all, lens);
    if (cap < 0) return -1;
    for (int r = 0; r < world; r++) ptr[r] = all + (size_t)r * (size_t)cap;
    GZ (gz_fastq_zip_finish (zf, ptr, lens, world));
    free (all);
    /* the finished VBlocks, one after the other in one device buffer, to the writer rank at their exact size */
    GZ (gz_fastq_zip_collect (zf, vbs, n_vbs, dev_z, z_cap, z_offsets));
    uint64_t mine = z_offsets[n_vbs], *d_len;
    CK (hipMalloc ((void **)&d_len, (size_t)(world + 1) * 8));
    CK (hipMemcpyAsync (d_len + world, &mine, 8, hipMemcpyHostToDevice, st));
    NC (ncclAllGather (d_len + world, d_len, 1, ncclUint64, comm, st));
    uint64_t all_len[64];
    CK (hipMemcpyAsync (all_len, d_len, (size_t)world * 8, hipMemcpyDeviceToHost, st));
    CK (hipStreamSynchronize (st));
    NC (ncclGroupStart ());
    if (rank != writer) NC (ncclSend (dev_z, mine, ncclUint8, writer, comm, st));
    else {
        uint64_t at = 0;
        for (int r = 0; r < world; r++) {
            if (r == writer) CK (hipMemcpyAsync (dev_gathered + at, dev_z, mine, hipMemcpyDeviceToDevice, st));
            else             NC (ncclRecv (dev_gathered + at, all_len[r], ncclUint8, r, comm, st));
            gathered_len[r] = all_len[r]; at += all_len[r];
        }
    }
    NC (ncclGroupEnd ());
    CK (hipStreamSynchronize (st));
    CK (hipFree (d_len));
    return 0;                                                      /* the writer hands dev_gathered to zfile_output_processed_vb_ext in vblock_i order */
}

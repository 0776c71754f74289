/*
 * mrx_comm.h -- results exchange between the ranks of a sharded batch, behind the C ABI.
 *
 * The reference is single-process; it has no counterpart.  Contract: SURVEY.md 8(e) -- texts are
 * sharded over the GPUs of a node by contiguous index ranges, the scan needs no collective, and
 * the ONE optional exchange step moves results only:
 *   match_first / search / captures  fixed bytes per text  -> plain all-gather
 *   findall (CSR spans)               variable              -> all-gather of per-rank {texts, spans},
 *                                                              then an all-gatherv into prefix-sum
 *                                                              offsets (RCCL has none: grouped
 *                                                              ncclBroadcast, one per rank)
 * Implemented on RCCL directly (librccl.so.1 is opened at the first mrx_comm_* call; a process that
 * never calls them needs no RCCL).  One process per GPU; every call is collective over the
 * communicator and enqueues on `stream` of the calling rank's current device.
 *
 * ONE STREAM PER COMMUNICATOR, CALLS SERIALISED: the size words and the staging of a communicator are single
 * buffers reused by every call, so two exchanges of one communicator must not be in flight at the same time
 * (enqueue them on one stream, or order the streams with events; use a second communicator for a second stream).
 *
 * ERRORS ARE COLLECTIVE: a rank whose own arguments are invalid, or whose output buffers are too small, still goes
 * through the size exchange, and EVERY rank then returns the same code (MRX_E_ARGUMENT / MRX_E_CAPACITY; padded
 * form: in *d_status) -- no rank is left waiting in a collective the others have abandoned.  Two things cannot be
 * told to the others and must hold on every rank: the arguments that size the collectives themselves
 * (cap_spans_per_rank, n_global, bytes_per_rank), and memory for the staging (mrx_comm_reserve at set-up rules
 * an allocation failure inside a call out).
 *
 * Host-language binding (Mojo): INTEGRATION.md, "Sharded batches".
 */
#ifndef MRX_COMM_H
#define MRX_COMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mrx_comm mrx_comm;

enum { MRX_COMM_ID_BYTES = 128 };   /* = NCCL_UNIQUE_ID_BYTES */

/* Rank 0 makes an id (ncclGetUniqueId) and hands its 128 bytes to the other ranks over any channel
 * the host has (a file, a socket, torch.distributed, MPI ...); every rank then joins. */
int mrx_comm_unique_id(uint8_t id[MRX_COMM_ID_BYTES]);
/* ncclCommInitRank on the calling thread's current HIP device.  Collective. */
int mrx_comm_init(const uint8_t id[MRX_COMM_ID_BYTES], int nranks, int rank, mrx_comm** out);
void mrx_comm_free(mrx_comm* c);
int mrx_comm_rank(const mrx_comm* c);
int mrx_comm_size(const mrx_comm* c);

/* Staging of the padded form of mrx_allgatherv_spans for a job of n_global texts and cap_spans_per_rank span slots;
 * mrx_comm_reserve(c, bytes) allocates it now.  A call that finds its staging too small grows it itself -- with a
 * device synchronisation and an allocation inside the call; reserving at set-up keeps both out of the timed path. */
size_t mrx_comm_spans_staging_bytes(const mrx_comm* c, int64_t n_global, int64_t cap_spans_per_rank);
int mrx_comm_reserve(mrx_comm* c, size_t bytes);

/* Fixed-size results: every rank contributes `bytes_per_rank` bytes (int32 start[n] / end[n] of
 * mrx_match_first_dev / mrx_search_dev, the span rows of mrx_captures_dev, uint8 flags ...);
 * d_recv[r * bytes_per_rank ...] = rank r's bytes, i.e. global text order for contiguous shards of
 * equal size.  One ncclAllGather.  No host synchronisation. */
int mrx_allgather_fixed(mrx_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream);

/* Rows of a fixed width, a different number per rank (shards that differ by a text, or any other
 * ragged split): rank r contributes rows_local rows of row_bytes bytes.  EXACT form: the per-rank
 * row counts are all-gathered (8 bytes each) and read back once (the only host synchronisation),
 * then every rank's rows travel by one grouped ncclBroadcast straight into their final place.
 * d_out receives all rows in rank order; *rows_total (host, may be NULL) their number;
 * MRX_E_CAPACITY on every rank when out_cap_rows is too small on ANY rank (nothing is written then). */
int mrx_allgatherv_rows(mrx_comm* c, const void* d_send, int64_t rows_local, size_t row_bytes,
                        void* d_out, int64_t out_cap_rows, int64_t* rows_total, void* stream);

/* findall results of all ranks as ONE CSR on every rank.
 *   in : d_prefix[n_local + 1], d_spans[.. d_prefix[n_local]][2]  -- what mrx_findall_*_dev wrote
 *   out: d_gprefix[N + 1] (N = all ranks' texts, rank order = global text order),
 *        d_gspans[T][2]   (T = all ranks' spans)
 * Steps: all-gather of {n_local, total_local} -> per-rank offsets; every rank
 * shifts its own prefix by the spans of the ranks before it on the device; grouped ncclBroadcast
 * of the shifted prefix segments and of the spans into their offsets.
 * cap_spans_per_rank == 0: EXACT form -- the 16 x nranks bytes of sizes are read back once so that
 *   the broadcasts carry exactly the bytes that exist (one host synchronisation per call).
 * cap_spans_per_rank  > 0: PADDED form -- no host synchronisation (given reserved staging, see above): the sizes stay on the
 *   device, every rank ships cap_spans_per_rank span slots and ceil(N / nranks) + 1 prefix slots
 *   (ncclAllGather into staging owned by the communicator), and a kernel compacts the staged rows
 *   into the global CSR.  n_local may differ between ranks by what a contiguous split leaves
 *   (rank r owns texts [r N / G, (r + 1) N / G)); a rank with more than cap_spans_per_rank spans
 *   sets the error word: *d_status (int32 on the device, may be NULL) becomes MRX_E_CAPACITY.
 * N_total / T_total (host, may be NULL): filled in the exact form only.
 * gprefix_cap / gspans_cap: capacities in entries / spans; when they do not hold the result on ANY rank, every rank
 * reports MRX_E_CAPACITY and nothing is written (exact form: the return value; padded form: *d_status). */
int mrx_allgatherv_spans(mrx_comm* c, const int64_t* d_prefix, int64_t n_local, const int32_t* d_spans,
                         int64_t cap_spans_per_rank, int64_t n_global,
                         int64_t* d_gprefix, int64_t gprefix_cap, int32_t* d_gspans, int64_t gspans_cap,
                         int64_t* N_total, int64_t* T_total, int32_t* d_status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRX_COMM_H */

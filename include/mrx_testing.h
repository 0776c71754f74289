/*
 * mrx_testing.h -- measurement and testing hooks of libmrx_hip.so.
 *
 * NOT part of the drop-in boundary (include/mrx.h): nothing here corresponds to a reference
 * interface.  bench.py uses the timing hooks for the roofline object; the parity tests use the
 * debug switches to run one batch through two kernel families and compare them text by text.
 * The switches are process-wide; set them while no call is in flight.
 */
#ifndef MRX_TESTING_H
#define MRX_TESTING_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Average duration (ms) of the dominant scan kernel over the launches made by
 * this thread since the last reset, measured with HIP events on the launch
 * stream; launches = number of scan-kernel launches measured. */
void mrx_timing_reset(void);
void mrx_timing_enable(int on);
double mrx_timing_scan_ms(int64_t* launches);
const char* mrx_last_kernel_name(void);
/* Testing aid: route every call to the generic lane-per-text kernels (the streaming kernel is
 * then never used) so that the two implementations can be compared on the same batch. */
void mrx_debug_force_generic(int on);
/* Testing aid: the treatments of long texts (pieces cut at synchronising bytes, a wavefront per text) are chosen by
 * the batch's average text length; 1 = always use them (pieces of 200 bytes where the plan has synchronising bytes),
 * 2 = never, 3 = as 1 but stepper plans take the wavefront-per-text kernel instead of pieces, 0 = by length. */
void mrx_debug_long_text_kernels(int mode);
/* findall of streamable plans runs as three launches (scan -> prefix sums -> decode; the default, and
 * the faster form as measured) or as ONE (scan, CSR offsets by decoupled look-back and spans fused);
 * same results either way.  0 = three launches, 1 = one launch when a 64-text task has at least
 * 32 KiB, 2 = one launch for short texts too.  Environment: MRX_FUSED=0|1|2 at compile time. */
void mrx_debug_fused_findall(int mode);
/* 1: findall of streamable plans over fixed-pitch texts of at most 1 KiB (16-byte aligned pitch, search automaton in
 * byte or code columns) runs as ONE launch that keeps every text's event bits in registers and writes offsets and
 * spans at their final place (k_stream_bits: no record stream, 1.27 GB of traffic instead of 1.74 GB on the headline
 * batch).  0 (default) = scan -> prefix sums -> decode: the one launch measured 4 % slower on 1 KiB texts and 1.5-3 x
 * slower on shorter ones (profiles/r04_stream_bits.md: its expansion costs as many VALU instructions as the scan).
 * Same results.  Environment: MRX_STREAM_BITS=1. */
void mrx_debug_stream_bits(int on);
/* Measurement: k_stream_bits writes four device clock readings (10 ns ticks) per 64-text task into d_trace
 * [4 * tasks]: task taken, texts walked and count published, count of all texts before known, spans written.
 * NULL = off (default). */
void mrx_debug_stream_bits_trace(int64_t* d_trace);
/* Ragged (CSR) batches of streamable plans with a reset byte are scanned by k_stream_dyn -- 256-text tasks,
 * a lane takes the next text when its own ends -- from 16384 texts up; 1 = always, 2 = never, 0 = by size. */
void mrx_debug_dynamic_texts(int mode);
/* 1: findall of a fixed-pitch batch of 2^18 texts and more runs as two halves on two streams (the decode of the
 * first under the scan of the second); 0 (default: the split measured slower) = one batch, three launches on the
 * caller's stream.  Results are the same. */
void mrx_debug_split_findall(int on);
/* findall of a batch full of matches by event rows (k_stream_findall<ST_ROWS> + k_decode_rows; texts of 2 KiB and more at
 * an aligned fixed pitch, common length a multiple of 128): 0 = when the handle's previous call found one match per 20
 * bytes or more (default), 1 = whenever the batch has the shape, 2 = never */
void mrx_debug_dense_rows(int mode);
/* PF_MW_TRIES plans: 1 = always the pending-tries walk; 0 (default) = the handle times it against
 * marks + stepper on the first calls of a batch shape (256 texts and more) and keeps the faster route. */
void mrx_debug_tries_always(int on);
/* regex.sub with \\1..\\9 on a deterministic chain: 1 = always measure every match's replacement
 * (k_subc_sizes), 0 (default) = templates under which every match gains the same number of bytes skip that pass. */
void mrx_debug_chain_sub_general(int on);
/* Host-side run of the one-pass table of an empty-match plan whose walks read beyond their match (build_emptywalk2(),
 * mrx_plan.cpp): findall of ONE text on the CPU, for tests that pin the table to the oracle without a GPU.  Returns the
 * number of spans (spans[2 k], spans[2 k + 1] for k < min(count, cap)), -1 when the handle has no such table. */
int mrx_testing_emptywalk_findall(const mrx_handle* h, const uint8_t* text, int len, int32_t* spans, int cap);
/* sub assembled from findall spans: lanes that share one text in k_subs_wave (16, 32 or 64; texts whose
 * frame or output exceed the group's LDS tiles go to k_subs_emit); 0 = k_subs_emit for every text,
 * anything else = chosen from the average text length. */
void mrx_debug_subs_group(int lanes);
/* The backtracking matcher's literal pass (k_litscan): 0 = one lane per text, 1 = always in pieces (208 bytes, so
 * that short test texts are cut, at positions that are not multiples of 16), anything else = in 2 KiB pieces
 * when the batch has few texts.  Results are the same. */
void mrx_debug_litscan_pieces(int mode);
/* Stepper plans with a multi-walk table (several walks side by side in one pass, k_mwalk; `multiwalk=yes` in
 * mrx_describe) use it for findall / count / search; 2 = never (the windowed stepper's restart-per-position loop
 * instead; also no backward marks and no fixed-length form of the bitset union pass), 3 = the multi-walk kernel
 * without its packed-start form (starts as 16-bit halves moved by byte permutes, texts below 64 KiB), anything else
 * = where the plan has one.  Results are the same. */
void mrx_debug_multiwalk(int mode);
/* Placement experiments (profiles/r03_scan_forms.md): the streaming findall's record stream begins `bytes` (a multiple
 * of 16) behind the start of its scratch allocation.  Results are the same. */
void mrx_debug_rec_skew(int64_t bytes);
/* include/mrx_comm.h, padded form of mrx_allgatherv_spans: its two device steps on buffers the caller fills as
 * ncclAllGather would have, so that the multi-rank arithmetic can be checked on one GPU.
 * meta_all: meta_stride int64 words per rank -- 2: {texts, spans}; 4: {texts (-1: that rank's arguments were invalid),
 * spans, capacity of that rank's global offsets buffer, of its global spans buffer}, the words the library gathers.
 * shift: out[i] = prefix[i + 1] + (spans of the ranks before `rank`) for i < n_local, 0 up to pad_to.
 * compact: stage_prefix[r][P], stage_spans[r][cap][2] -> global CSR; *d_status = MRX_OK, MRX_E_CAPACITY (some rank's
 * capacities, or these, do not hold the result: nothing is written) or MRX_E_ARGUMENT (some rank was invalid). */
int mrx_testing_comm_shift(const int64_t* d_prefix, int64_t n_local, const int64_t* d_meta_all, int meta_stride, int rank,
                           int64_t* d_out, int64_t pad_to, void* stream);
int mrx_testing_comm_compact(const int64_t* d_meta_all, int meta_stride, int nranks, const int64_t* d_stage_prefix, int64_t P,
                             const int32_t* d_stage_spans, int64_t cap, int64_t* d_gprefix, int64_t gprefix_cap,
                             int32_t* d_gspans, int64_t gspans_cap, int32_t* d_status, void* stream);
/* Bytes of device memory the calling thread's scratch arenas hold (see mrx_release_scratch). */
size_t mrx_debug_scratch_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* MRX_TESTING_H */

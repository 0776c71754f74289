// Internal glue between the translation units of libmrx_hip.so (not part of any ABI).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace mrx {
// records the calling thread's last error message (mrx_last_error()) and returns `code`
int internal_fail(int code, const std::string& msg);

// ---- mrx_stream_bits.hip: findall of short fixed-pitch texts in one launch, events kept in registers ----------
struct DevPlan;
// 0 (default) / 1: MRX_STREAM_BITS=1 in the environment, mrx_debug_stream_bits(1) at run time
int stream_bits_mode();
void stream_bits_set_mode(int on);
void stream_bits_set_trace(int64_t* d_trace);   // measurement: 4 x int64 per 64-text task (nullptr = off)
// can this plan / batch shape take the form? (fixed pitch, 16-byte aligned, texts of at most 1 KiB, a search
// automaton in byte or code columns)
bool stream_bits_eligible(const DevPlan& p, const uint8_t* data, int64_t stride, int64_t max_len, int64_t n);
size_t stream_bits_ctrl_words(int64_t n);   // 8-byte words of look-back state the launch needs (zeroed by it)
size_t stream_bits_args_bytes();            // bytes of device memory for its argument block
// init: zeroes the look-back words and writes the argument block; scan: the one launch.  Both on `stream`.
// d_prefix[n + 1], d_spans[span_cap][2], d_total are the outputs.
int stream_bits_init(int64_t n, int64_t max_len, int64_t* d_prefix, int32_t* d_spans, int64_t span_cap, int64_t* d_total, void* d_ctrl,
                     void* d_args, void* stream);
int stream_bits_scan(const DevPlan& p, const uint8_t* d_blob, const uint8_t* data, int64_t stride, const int32_t* lens,
                     int32_t len, int64_t max_len, int64_t n, const void* d_args, void* stream);
}  // namespace mrx

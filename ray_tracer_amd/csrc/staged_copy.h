// Large host-generated tables go to the device through two pinned pieces: host threads write one piece while the other one
// crosses the link.  (Built whole in a std::vector first, a table pays for zero-filling it, for filling it again, and for a copy
// out of pageable memory at a fifth of the link's rate: the tables of a 10 M-triangle scene took 0.5 s that way.)
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstddef>
#include <cstring>
#include <memory>
#include "parallel_for.h"

namespace mi355rt {

struct PinnedPieces {
	static constexpr size_t kBytes = 32u << 20; // per piece
	void *buf[2] = {nullptr, nullptr};
	hipEvent_t done[2] = {nullptr, nullptr};

	hipError_t ensure()
	{
		for (int k = 0; k < 2; k++) {
			if (buf[k]) continue;
			hipError_t e = hipHostMalloc(&buf[k], kBytes, hipHostMallocDefault);
			if (e != hipSuccess) { buf[k] = nullptr; return e; }
			e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming);
			if (e != hipSuccess) { (void)hipHostFree(buf[k]); buf[k] = nullptr; return e; }
		}
		return hipSuccess;
	}
	void release()
	{
		for (int k = 0; k < 2; k++)
			if (buf[k]) { (void)hipHostFree(buf[k]); (void)hipEventDestroy(done[k]); buf[k] = nullptr; done[k] = nullptr; }
	}
};

// dst[0 .. n_items) on the device, item_bytes each; fill(begin, end, out) writes items [begin, end) to out (host memory, item
// `begin` first) and is called from several threads on disjoint ranges.  Returns when the last piece has arrived.
template <class Fill>
hipError_t upload_generated(PinnedPieces &pp, void *dst, size_t n_items, size_t item_bytes, hipStream_t stream, Fill fill)
{
	if (n_items == 0) return hipSuccess;
	if (n_items * item_bytes <= (4u << 20)) { // a small table: not worth pinned memory
		std::unique_ptr<char[]> whole(new char[n_items * item_bytes]);
		fill((size_t)0, n_items, whole.get());
		const hipError_t c = hipMemcpyAsync(dst, whole.get(), n_items * item_bytes, hipMemcpyHostToDevice, stream);
		const hipError_t w = hipStreamSynchronize(stream);
		return c != hipSuccess ? c : w;
	}
	hipError_t e = pp.ensure();
	if (e != hipSuccess) return e;
	const size_t per_piece = std::max<size_t>(PinnedPieces::kBytes / item_bytes, 1);
	const size_t n_pieces = (n_items + per_piece - 1) / per_piece;
	for (size_t c = 0; c < n_pieces && e == hipSuccess; c++) {
		const int k = (int)(c & 1);
		if (c >= 2) e = hipEventSynchronize(pp.done[k]); // the copy that read this piece last
		if (e != hipSuccess) break;
		const size_t at = c * per_piece, m = std::min(per_piece, n_items - at);
		char *piece = static_cast<char *>(pp.buf[k]);
		parallel_for(m, 1 << 13, [&](uint64_t b, uint64_t t) { fill(at + b, at + t, piece + b * item_bytes); });
		e = hipMemcpyAsync(static_cast<char *>(dst) + at * item_bytes, piece, m * item_bytes, hipMemcpyHostToDevice, stream);
		if (e == hipSuccess) e = hipEventRecord(pp.done[k], stream);
	}
	const hipError_t s = hipStreamSynchronize(stream); // also after an error: nothing may still read the pieces when they are reused
	return e != hipSuccess ? e : s;
}

// bytes from the device into pageable host memory: piece c + 1 crosses the link while host threads copy piece c out of its pinned
// buffer (a plain hipMemcpy into pageable memory runs at a fifth of the link's rate).  Returns when dst is complete.
inline hipError_t download_staged(PinnedPieces &pp, const void *src, size_t bytes, void *dst, hipStream_t stream)
{
	if (bytes == 0) return hipSuccess;
	if (bytes <= (4u << 20)) {
		const hipError_t c = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
		const hipError_t w = hipStreamSynchronize(stream);
		return c != hipSuccess ? c : w;
	}
	hipError_t e = pp.ensure();
	if (e != hipSuccess) return e;
	const size_t n_pieces = (bytes + PinnedPieces::kBytes - 1) / PinnedPieces::kBytes;
	auto request = [&](size_t c) -> hipError_t {
		const size_t at = c * PinnedPieces::kBytes, m = std::min(PinnedPieces::kBytes, bytes - at);
		const hipError_t r = hipMemcpyAsync(pp.buf[c & 1], static_cast<const char *>(src) + at, m, hipMemcpyDeviceToHost, stream);
		return r == hipSuccess ? hipEventRecord(pp.done[c & 1], stream) : r;
	};
	e = request(0);
	for (size_t c = 0; c < n_pieces && e == hipSuccess; c++) {
		e = hipEventSynchronize(pp.done[c & 1]);
		if (e != hipSuccess) break;
		const size_t at = c * PinnedPieces::kBytes, m = std::min(PinnedPieces::kBytes, bytes - at);
		const char *from = static_cast<const char *>(pp.buf[c & 1]);
		char *to = static_cast<char *>(dst) + at;
		// the next piece may only be requested into the other buffer, which the copy below does not touch
		if (c + 1 < n_pieces) e = request(c + 1);
		parallel_for(m, 1 << 20, [&](uint64_t b, uint64_t t) { std::memcpy(to + b, from + b, t - b); });
	}
	const hipError_t s = hipStreamSynchronize(stream);
	return e != hipSuccess ? e : s;
}

} // namespace mi355rt

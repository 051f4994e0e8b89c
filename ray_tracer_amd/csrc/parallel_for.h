// Splits a loop over [0, n) among host threads: the per-triangle and per-polygon tables of a scene are independent entries.
#pragma once
#include <algorithm>
#include <cstdint>
#include <system_error>
#include <thread>
#include <vector>

namespace mi355rt {

// body(begin, end) is called once per thread on a contiguous part and must not throw; small loops run on the calling thread
template <class F>
void parallel_for(uint64_t n, uint64_t grain, F body)
{
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	const uint64_t parts = std::min<uint64_t>(std::min(hw, 16u), grain ? n / grain : 1);
	if (parts < 2) { body((uint64_t)0, n); return; }
	std::vector<std::thread> pool;
	pool.reserve(parts - 1);
	uint64_t started = 1;
	try {
		for (; started < parts; started++) pool.emplace_back([=] { body(n * started / parts, n * (started + 1) / parts); });
	} catch (const std::system_error &) {
		// no more threads to be had: the calling thread takes the parts that are left
	}
	body((uint64_t)0, n / parts);
	for (uint64_t p = started; p < parts; p++) body(n * p / parts, n * (p + 1) / parts);
	for (auto &t : pool) t.join();
}

} // namespace mi355rt

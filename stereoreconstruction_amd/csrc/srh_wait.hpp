// srh_wait.hpp -- a wait that cannot last for ever.  Everything in the multi-GPU exchange that waits for a peer goes
// through bounded_wait(): the rendezvous and the enqueue calls of a non-blocking RCCL communicator (srh_comm.hip:
// rccl_settle) AND the completion of a collective already on a stream (rccl_wait_stream: a peer that dies INSIDE the
// collective never completes it -- a bare hipStreamSynchronize would hold the rank for ever).
// Plain C++ (no HIP, no RCCL): tests/sharded_host_test.cpp drives it on the CPU with a transport that never completes.
#pragma once

#include <chrono>
#include <cstdio>
#include <functional>
#include <thread>

namespace srh {

// poll():  0 = complete, 1 = still in progress, anything else = failed (`error()` then gives the text, may be null).
// Returns nullptr when complete, else an error text (the timeout's is written to msg).  Polls every 200 us.
inline const char *bounded_wait(const std::function<int()> &poll, const std::function<const char *()> &error,
                                int timeout_ms, const char *what, char *msg, size_t msg_len)
{
	const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
	for (;;) {
		const int st = poll();
		if (st == 0) return nullptr;
		if (st != 1) {
			const char *e = error ? error() : nullptr;
			if (e) return e;
			snprintf(msg, msg_len, "%s failed", what);
			return msg;
		}
		if (std::chrono::steady_clock::now() > t_end) {
			snprintf(msg, msg_len, "%s still in progress after %d ms (a rank missing or gone?)", what, timeout_ms);
			return msg;
		}
		std::this_thread::sleep_for(std::chrono::microseconds(200));
	}
}

} // namespace srh

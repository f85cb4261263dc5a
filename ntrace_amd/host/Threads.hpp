// Threads.hpp -- exception-safe fork / join for the host builder's worker threads.
//
// A bare std::thread whose function throws (std::bad_alloc, FatalError) calls std::terminate, and destroying a joinable
// std::thread -- which is what happens to the earlier ones when a later std::thread constructor throws std::system_error --
// terminates too.  ThreadGroup runs every task under a catch-all that keeps the first exception, runs a task INLINE when no thread
// can be started, always joins (also from its destructor), and rethrows the kept exception on the calling thread in join(), from
// where it reaches the try / catch of the C-ABI wrapper (host_api.cpp).
#pragma once
#include <exception>
#include <mutex>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

namespace FW {

class ThreadGroup {
public:
    ThreadGroup() {}
    ~ThreadGroup() { joinAll(); }
    ThreadGroup(const ThreadGroup&) = delete;
    ThreadGroup& operator=(const ThreadGroup&) = delete;

    template <class F>
    void spawn(F f)
    {
        auto guarded = [this, f]() mutable {
            try { f(); } catch (...) { keep(std::current_exception()); }
        };
        try {
            m_threads.emplace_back(guarded);
        } catch (const std::system_error&) {   // thread / pid limit: do the work here instead
            guarded();
        }
    }

    // the calling thread's own share of the work, under the same guard
    template <class F>
    void run(F f)
    {
        try { f(); } catch (...) { keep(std::current_exception()); }
    }

    void join()
    {
        joinAll();
        if (m_error) {
            std::exception_ptr e = m_error;
            m_error = nullptr;
            std::rethrow_exception(e);
        }
    }

private:
    void joinAll()
    {
        for (std::thread& t : m_threads)
            if (t.joinable()) t.join();
        m_threads.clear();
    }
    void keep(std::exception_ptr e)
    {
        std::lock_guard<std::mutex> lk(m_mu);
        if (!m_error) m_error = e;
    }
    std::vector<std::thread> m_threads;
    std::mutex m_mu;
    std::exception_ptr m_error;
};

}  // namespace FW

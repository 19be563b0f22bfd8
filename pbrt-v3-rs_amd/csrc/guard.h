// Nothing C++ leaves the library through its C ABI ("no C++ exception or abort may cross the boundary"; the reference's own failure mode on the same
// conditions is a panic).  Every extern "C" entry point runs its body inside ph_guard: std::bad_alloc becomes PBRT_HIP_ERR_OOM, anything else PBRT_HIP_ERR_DEVICE, with
// pbrt_hip_last_error saying what was caught.  Helper threads run through ThreadGroup: a thread that cannot be started runs its work on the caller instead, a worker's
// exception is carried to the caller's thread (where the guard turns it into a status code), and no thread outlives the call that started it.
#pragma once
#include "../../include/pbrt_hip.h"
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

struct PbrtHipScene;
namespace phost {
int set_err(PbrtHipScene* s, int code, const std::string& msg);

inline int caught(PbrtHipScene* s, int code, const char* where, const char* what) noexcept {
    try { return set_err(s, code, std::string(where) + ": " + what); } catch (...) { return code; }   // (no memory even for the message: the code alone)
}

template <class F> inline int ph_guard(PbrtHipScene* s, const char* where, F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { return caught(s, PBRT_HIP_ERR_OOM, where, "out of host memory (std::bad_alloc)"); }
    catch (const std::exception& e) { return caught(s, PBRT_HIP_ERR_DEVICE, where, e.what()); }
    catch (...) { return caught(s, PBRT_HIP_ERR_DEVICE, where, "unknown C++ exception"); }
}
template <class T, class F> inline T* ph_guard_ptr(const char* where, F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { (void)caught(nullptr, PBRT_HIP_ERR_OOM, where, "out of host memory (std::bad_alloc)"); }
    catch (const std::exception& e) { (void)caught(nullptr, PBRT_HIP_ERR_DEVICE, where, e.what()); }
    catch (...) { (void)caught(nullptr, PBRT_HIP_ERR_DEVICE, where, "unknown C++ exception"); }
    return nullptr;
}
template <class F> inline void ph_guard_void(F&& body) noexcept { try { body(); } catch (...) {} }

class ThreadGroup {
public:
    ThreadGroup() = default;
    ThreadGroup(const ThreadGroup&) = delete;
    ThreadGroup& operator=(const ThreadGroup&) = delete;
    ~ThreadGroup() { for (std::thread& t : th_) if (t.joinable()) t.join(); }
    // f() on a new thread — or right here when the system has no thread to give (std::system_error) or no memory for the bookkeeping
    template <class F> void run(F f) {
        auto body = [this, f]() mutable {
            try { f(); } catch (...) { std::lock_guard<std::mutex> g(mu_); if (!first_) first_ = std::current_exception(); }
        };
        bool started = false;
        try { th_.emplace_back(body); started = true; } catch (const std::system_error&) {} catch (const std::bad_alloc&) {}
        if (!started) body();
    }
    // waits for every thread; rethrows the first exception a worker met
    void join() {
        for (std::thread& t : th_) if (t.joinable()) t.join();
        th_.clear();
        if (first_) { std::exception_ptr e = first_; first_ = nullptr; std::rethrow_exception(e); }
    }
private:
    std::vector<std::thread> th_;
    std::exception_ptr first_;
    std::mutex mu_;
};

}  // namespace phost

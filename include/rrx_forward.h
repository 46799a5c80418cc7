/* Glue between the reference-style C++ API and the C ABI of librrx_hip.so: status -> exception (the reference
 * throws std::runtime_error / prints and `throw 1`, include/tools_gpu.h:22-29), and the per-thread launch stream. */
#ifndef RRX_FORWARD_H
#define RRX_FORWARD_H
#include <stdexcept>
#include <string>
#include "rrx_hip.h"
#include "types.h"

namespace rrx_host
{
    inline void*& current_stream() { static thread_local void* s = nullptr; return s; }   // nullptr = default stream
    inline void set_stream(void* s) { current_stream() = s; }
    inline void check(const int status)
    {
        if (status != 0)
            throw std::runtime_error(std::string("rrx: ") + rrx_last_error());
    }
}
#define RRX_CALL(name, ...) rrx_host::check(RRX_SFX(name)(__VA_ARGS__, rrx_host::current_stream()))
#endif

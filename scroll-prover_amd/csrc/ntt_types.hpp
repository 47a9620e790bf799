// ntt_types.hpp -- the twiddle-table descriptor of the 29-bit NTT kernels (ntt29.hpp), split out so that the host-side plan cache
// (lib_common.hpp) can hold it without pulling in the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zk {
struct Tw29 { const uint4 *lo; const uint4 *hi; const uint32_t *top; };   // entry i: limbs 0-3, 4-7, 8 of w^i * 2^261 mod r
}  // namespace zk

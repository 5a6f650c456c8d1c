// rbd_jit.hpp — run-time specialisation (rbd_jit.hip): source generator, hiprtc compile with an on-disk cache.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rbd_state_plan.hpp"

namespace rbd {
bool jit_available();  // libhiprtc found and RBD_JIT != 0
// dtype: RBD_F32 / RBD_F64 — the kernels of one scalar type per program (the dense step's kernels only in fp32, nv a multiple of 4, nv <= 40)
std::string spec_source(const StatePlan& P, int nb, int nq, int nv, const uint64_t* row_mask, const double* gravity, int dtype);
std::vector<char> jit_code_object(const std::string& source, std::string* log);
}  // namespace rbd

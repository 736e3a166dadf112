// Compiles one kernel source file for one 16-bit operand type (common.h "OPERAND TYPE"):
//   hipcc -DSABER_OP_NS=op_bf16              -DSABER_OP_SRC='"gemm.hip"' -c op_wrap.hip -o build/bf16/gemm.o
//   hipcc -DSABER_OP_NS=op_f16 -DSABER_OP_F16 -DSABER_OP_SRC='"gemm.hip"' -c op_wrap.hip -o build/f16/gemm.o
// Everything the source defines (kernels, launchers, file-level state) lands in that namespace, so the two objects link side by side;
// the headers below are included first, outside it (they carry #pragma once / include guards, the source's own #includes become no-ops).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <utility>
#include "common.h"
#include "kernels.h"

namespace SABER_OP_NS {
#include SABER_OP_SRC
}

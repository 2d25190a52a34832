// oracle/ref_shim (TEST INFRASTRUCTURE ONLY): common_host.h includes tcnn's gpu_matrix.h; nothing of it is used by the files compiled against this shim
#pragma once
#include <tiny-cuda-nn/common.h>

// oracle/ref_shim: see tiny-cuda-nn/common.h in this directory
#pragma once
#include <tiny-cuda-nn/common.h>

// Private: entry points shared between gemm.hip (dispatcher) and gemm_big.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "g2vlm_hip.h"

bool g2v_gemm_big_eligible(const g2v_gemm_desc* d);
int g2v_gemm_big_launch(const g2v_gemm_desc* d, hipStream_t s);
bool g2v_gemm_8p_supported(const g2v_gemm_desc* d);
bool g2v_gemm_8p_preferred(const g2v_gemm_desc* d);
int g2v_gemm_8p_launch(const g2v_gemm_desc* d, hipStream_t s);
// gemm_4w.hip: the four-wave form of the same tile; bm / order (large group first) as chosen by g2v_gemm_8p_launch
int g2v_gemm_4w_launch(const g2v_gemm_desc* d, int bm, const int* order, hipStream_t s);
bool g2v_gemm_skinny_eligible(const g2v_gemm_desc* d);
int g2v_gemm_skinny_launch(const g2v_gemm_desc* d, hipStream_t s);

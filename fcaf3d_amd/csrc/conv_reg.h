// Internal (not exported) launchers of the register-direct convolution kernels (conv_reg.hip), called by the
// C-ABI dispatchers in conv.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum {
  FC_REG_64x64_SPLIT = 1,   // workgroup = one 64x64 tile, 4 waves split the reduction (LDS fixed-order sum)
  FC_REG_64x64_WAVE = 2,    // workgroup = 2x2 tiles of 64x64, one per wave, no LDS / barrier
  FC_REG_32x128_SPLIT = 3,
  FC_REG_64x128_SPLIT = 4,
  FC_REG_32x64_SPLIT = 5,
};

static inline void fc_reg_tile(int variant, int* rows, int* cols) {
  switch (variant) {
    case FC_REG_64x64_WAVE: *rows = 128; *cols = 128; break;     // per workgroup
    case FC_REG_32x128_SPLIT: *rows = 32; *cols = 128; break;
    case FC_REG_64x128_SPLIT: *rows = 64; *cols = 128; break;
    case FC_REG_32x64_SPLIT: *rows = 32; *cols = 64; break;
    default: *rows = 64; *cols = 64; break;
  }
}

// dst: `out` (S == 1) or the split workspace; cnt != NULL: pair mode (grid.z = offset, see k_conv_reg)
int fc_conv_reg_launch(const float* in, const float* W, const int* nbr, const int* out_index, const int* cnt, float* dst,
                       int64_t n_rows, int K, int Cin, int Cout, int S, int variant, hipStream_t stream);
// part: (S, K, Cin, Cout) partial gradients; cnt != NULL: exact pair lists, else dense table (pin = nbr or NULL)
int fc_wgrad_reg_launch(const float* in, const float* gout, const int* pin, const int* pout, const int* cnt, float* part,
                        int64_t n_out, int K, int Cin, int Cout, int S, int64_t rows_per_split, hipStream_t stream);

// streaming (persistent-wave) forward / backward-data kernel: see k_conv_stream
void fc_conv_stream_plan(int64_t n_rows, int K, int Cin, int Cout, int tile, int force_S, int* tm, int* S, int* items);
int fc_conv_stream_launch(const float* in, const float* W, const int* nbr, const unsigned int* gmask, const int* out_index,
                          float* dst, int64_t n_rows, int K, int Cin, int Cout, int tm, int S, hipStream_t stream);
int fc_group_masks_launch(const int* nbr, int64_t n_out, int K, unsigned int* gmask, hipStream_t stream);

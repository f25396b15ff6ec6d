// Compile-time-specialised circle-FFT passes (fft_fixed.hip) for the tile shapes the prover's committed columns
// actually have; the generic k_fft_staged in kernels_fft.hip covers every other shape.
#pragma once
#include "kernels.h"

namespace lmn {

// Layers [lo, lo + rbits) of a 2^log_n-point transform on tiles of 2^rbits rows x 2^cb contiguous words, in place or
// out of place.  src holds all 2^log_n words, or - zero_extended_top, the top pass of a forward transform - only the lower
// half (the coefficients of a polynomial evaluated on the domain of twice its size).  scale_log != 0: the inverse transform's outputs are
// multiplied by 2^scale_log (a 31-bit rotation).  Returns false when no instantiation covers the shape.
bool launch_fft_fixed_pass(bool inverse, uint32_t* data, uint64_t col_stride, const uint32_t* src, uint64_t src_stride, int lo,
                           int rbits, int cb, int log_n, const TwPtrs& tw, uint32_t scale_log, int ncols, int cpb,
                           uint32_t h_off, int xcd_swizzle, bool zero_extended_top, lmn_stream_t s);
// The fused strided pass of "interpolate, then evaluate on the domain of twice the size" (k_fft_interp_extend's role) for
// 2^log_n-point columns: coeffs hold the output of the inverse low pass, lde receives both halves.  false: unsupported size.
bool launch_interp_extend_fixed(uint32_t* coeffs, uint64_t coeff_stride, uint32_t* lde, uint64_t lde_stride, int log_n,
                                const TwPtrs& itw, const TwPtrs& tw_ext, int ncols, lmn_stream_t s);
// The AoS -> SoA transpose (padding rows, canonical-word check: launch_transpose_pad's contract) fused into the first pass
// of the interpolation of 2^log_n-row columns: `rows` = the table's n_rows x ncols words; coeffs receives what
// launch_fft_fixed_pass(inverse, lo 0, 12 layers) would have produced from the transposed columns.  false: not applicable.
bool launch_fft_rows_fixed(uint32_t* coeffs, uint64_t col_stride, const uint32_t* rows, uint64_t n_rows, int ncols, int log_n,
                           const PadRow& pad, uint32_t* bad_flag, uint32_t bad_value, const TwPtrs& itw, lmn_stream_t s);

}  // namespace lmn

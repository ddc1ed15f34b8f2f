// trx_tables.h -- layout of the device-resident table blob shared by host table generation
// (trx_tables.cpp) and the gfx950 kernels (trx_kernels.hip).
//
// The blob is what sigProcLibSetup() builds in the reference (Transceiver52M/sigProcLib.cpp:2139-2172):
// decimator taps, 64 fractional-delay filters, the 1-SPS reverse GMSK rotation, all correlation
// sequences with gain/toa, plus two derived tables that let the kernels avoid fp64 and divisions:
//   * sincv[]   : sinc(M_PI_F * q/512) for q in [0,4096) -- every value interpolatePoint()
//                 (sigProcLib.cpp:1100-1118) can request, because peakDetect() only visits positions
//                 that are multiples of 1/512 (sigProcLib.cpp:1159-1175); XOR-swizzled for LDS banks
//   * gain_inv, ci_den per sequence (Complex.h:144-150 inv(); sigProcLib.cpp:1629)
// ~23 KB, generated once on the host (rank 0), broadcast over RCCL, resident in HBM/L2.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct trx_c32 { float re, im; };

// The fused demodulator evaluates composite taps u = TRX_FUSED_U0 .. TRX_FUSED_U0 + TRX_FUSED_NT - 1 of the 35 (trx_kernel4.hip,
// fir24x3): 6 .. 29 since round 5 (rounds 2-4: 8 .. 31 -- but taps 30 / 31 carry 3.7e-7 / 2e-9 of a filter whose taps sum to 1 and
// taps 6 / 7 carry 6.2e-7 / 3e-8: the same 24 multiply-adds two taps lower halve the error, profiles/r05_fused_taps.txt).
// Measurement builds (tools/build_variants.py name:all:-DTRX_FUSED_U0=..,-DTRX_FUSED_NT=..) override them to price other windows.
// Taps are fetched four at a time from 16-byte aligned LDS addresses: the kernel's LDS copy of a composite row is shifted by
// TRX_FUSED_SH so that tap U0 sits at a multiple of 4.
#ifndef TRX_FUSED_U0
#define TRX_FUSED_U0 6
#endif
#ifndef TRX_FUSED_NT
#define TRX_FUSED_NT 24
#endif
#define TRX_FUSED_SH ((4 - TRX_FUSED_U0 % 4) % 4)          /* shift of the LDS copy of a composite row */
#define TRX_FUSED_NTP ((TRX_FUSED_NT + 3) / 4 * 4)        /* taps per edge8 row (rows are fetched as float4) */

enum {
	TRX_SEQ_TSC0   = 0,   // gMidambles[0..7]
	TRX_SEQ_RACH0  = 8,   // gRACHSequences[0..2]
	TRX_SEQ_DUMMY  = 11,  // gDummySequence
	TRX_SEQ_EDGE0  = 12,  // gEdgeMidambles[0..7]
	TRX_SEQ_SCH    = 20,  // gSCHSequence
	TRX_NSEQ       = 21,
	TRX_SEQ_MAXLEN = 64,
	TRX_SINCV_LEN  = 4096,
	TRX_DELAY_FILTS = 64,
	TRX_DELAY_HLEN = 20,
};

struct trx_seq {
	trx_c32 taps[TRX_SEQ_MAXLEN];  // conjugated rotated +-1 sequence (CorrelationSequence::sequence)
	trx_c32 gain;                  // CorrelationSequence::gain
	trx_c32 gain_inv;              // gain.inv()
	float   ci_den;                // (N-1) * gain.abs()
	float   toa;                   // CorrelationSequence::toa
	int32_t n;                     // sequence length: 16 / 40 / 64
	float   ci_den_inv;            // 1 / ci_den
};

struct trx_tables {
	uint32_t magic;                // 'TRXT'
	uint32_t version;
	float    dec_taps[16];                              // Resampler(1,4) partition 0 (reversed)
	float    delay_filt[TRX_DELAY_FILTS][TRX_DELAY_HLEN];
	trx_c32  rrot1[160];                                // GMSKReverseRotation1 (157 used)
	float    c0_inv[8];                                 // EDGE equaliser, 5 taps used
	trx_seq  seq[TRX_NSEQ];
	float    sincv[TRX_SINCV_LEN];                      // index q ^ ((q >> 4) & 31)
	float    chan_taps[4][16];                          // Channelizer(4,*,16) sub-filters (reversed)
	float    rs6548_taps[65][16];                       // Resampler(65,48) partitions (reversed)
	// composite of fractional-delay filter f and the /4 decimator (fused demod): comp[f][u] = sum_{t+k=u} g[t]*h_f[k],
	// u < 35; row 64 = no fractional filter (|frac| <= 0.01, sigProcLib.cpp:1056): g shifted by 9
	float    comp_filt[TRX_DELAY_FILTS + 1][36];
	// EDGE 8-PSK demodulator constants (sigProcLib.cpp:691-711, :1962-2006, :2074-2093), evaluated with the host libm
	trx_c32  edge_derot[16];                            // (cosf(p), -sinf(p)), p = (float)(i%16)*3.0f*M_PI/8.0f
	trx_c32  edge_ideal[9];                             // (cos(ph), sin(ph)), ph = step*k, k = -4..4, step = 2*M_PI_F/8
	trx_c32  edge_rot2[2];                              // rotateBurst2 phasors for -M_PI/8 and -M_PI/4
	float    edge_step;                                 // 2.0f * M_PI_F / 8.0f
	float    edge_pad;
	// unit structure of the GMSK correlation sequences (trx_device.h, corr_unit()): bit s of unit_ok is set when every
	// tap k of seq[s] is (+-1, e) for even k / (e, +-1) for odd k with |e| <= 1e-13; bit k of unit_neg[s] = that +-1 is -1
	uint64_t unit_neg[TRX_NSEQ];
	uint32_t unit_ok;
	uint32_t unit_pad;
	// fused demodulator, low-side partial outputs: the reference's decimator starts on zero history, so output i only
	// sees the delayed samples n >= n_lo, i.e. decimator taps t >= t0 = n_lo + 15 - 4i (1..15).  Composite of delay
	// filter f with the decimator truncated to t >= t0:  edge_lo[f][t0-1][u] = sum_{t>=t0, t+k=u} g[t]*h_f[k], u < 35
	float    edge_lo[TRX_DELAY_FILTS + 1][15][36];
	// The same four rows re-packed for the usual geometry (n_lo = 0: outputs 0..3 see decimator taps t >= 15, 11, 7, 3),
	// in the form the main filter loop of the 4-SPS kernel consumes -- 24 taps per lane, tap u = U0 .. U0 + 23 of the lane's row (U0 = TRX_FUSED_U0):
	//   edge8[f][i][0..23]     = edge_lo[f][14 - 4i][U0 .. U0+23]        main part of output i (lanes 52..55)
	//   edge8[f][4 + i][k]     = edge_lo[f][14 - 4i][k - (8 - U0)]       its taps u < U0 (zero elsewhere), for a lane whose window starts 8
	//                                                                     samples early (lanes 56..59); taps u >= 32 are 0
	// (TRX_FUSED_U0 / TRX_FUSED_NT, below: the window of composite taps the fused demodulator runs; 8 / 24 in the text above)
	uint32_t fused_u0, fused_nt;                        // TRX_FUSED_U0 / TRX_FUSED_NT the composite rows (comp_filt's LDS shift, edge8) were generated
	                                                    // for: a kernel built for another window refuses the blob (and edge8 stays at a multiple of 16 bytes)
	float    edge8[TRX_DELAY_FILTS + 1][8][TRX_FUSED_NTP];
	// high-side partial outputs (a burst delayed by a negative whole shift w <= -2 ends at delayed sample n_hi = L - 1 + w:
	// output i only sees decimator taps t <= tm = n_hi + 15 - 4i, 0..14).  Composite of delay filter f with the decimator
	// truncated to t <= tm:  edge_hi[f][tm][u] = sum_{t<=tm, t+k=u} g[t]*h_f[k]  (u <= tm + 19; 0 beyond)
	float    edge_hi[TRX_DELAY_FILTS + 1][15][36];
};
static_assert(offsetof(trx_tables, edge8) % 16 == 0, "edge8 rows are fetched as float4");
static_assert(TRX_FUSED_U0 >= 0 && TRX_FUSED_U0 <= 8 && TRX_FUSED_U0 + TRX_FUSED_NTP + TRX_FUSED_SH <= 36 && TRX_FUSED_NTP <= 32,
	      "fused demodulator tap window");

#define TRX_TABLES_MAGIC   0x54585254u
#define TRX_TABLES_VERSION 8u   /* 8: edge8 rows for taps U0 = 6 .. 29 (round 5), window recorded in fused_u0 / fused_nt */

// XOR swizzle of the sincv index: conflict-free LDS gathers both for lanes whose positions differ
// by multiples of 16/512 (coarse bisection levels) and by 1/512 steps (fine levels).
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
int trx_sincv_swz(int q) { return q ^ ((q >> 4) & 31); }

// Host-side generation (no GPU needed).  Returns 0 on success.
int trx_tables_generate(trx_tables *out);

// Resampler::initFilters (Resampler.cpp:47-96) for an arbitrary rational ratio: out[path * filt_len + k], reversed taps
void trx_polyphase_taps(unsigned p, unsigned q, unsigned filt_len, float bw, float *out);

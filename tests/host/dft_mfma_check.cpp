// Host check of os2d_amd/csrc/dft_mfma.h on the SPMD emulator (spmd_emu.h): the forward and the inverse transform kernels -
// the same source the GPU runs - against float64 DFTs, for an untiled fast-path map, a map whose width is not a multiple of 4,
// a small map and tiled maps (ragged tiles).  Built and run by tests/test_dft_mfma_host.py.   usage: dft_mfma_check [H W C NB]...
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "spmd_emu.h"

#define OS2D_HOST_EMU 1
static int emu_dft_policy = 0;      // 0: smallest transform per map, 1: the canonical sizes
#define DFT_DEV static inline
#define DFT_TID emu::tid()
#define DFT_BID emu::bid()
#define DFT_GRID emu::grid()
#define DFT_LDS emu::lds()
#define DFT_BARRIER() emu::group_barrier()
#define DFT_MFMA(a, b, c) emu::mfma_32x32x16_f16(a, b, c)
#define DFT_SHFL_XOR(v, m) emu::shfl_xor(v, m)
#define DFT_SHFL_XOR_U32(v, m) emu::shfl_xor_u32(v, m)
#define DFT_BALLOT(p) emu::ballot(p)
#define DFT_FLAG int*
#define DFT_FLAG_SET(f) ((f) != nullptr)
#define DFT_RAISE(p) (*(p) = 1)
#define DFT_UNIFORM(x) (x)
#include "dft_mfma.h"

// The spectra layouts, stated independently of the kernels' helper (include/os2d_hip.h): blocks of 64 pairs,
// [pair / 64][quad of bins][pair % 64 (last block: what is left)][channel][4 bins][re | im]
static size_t spec_index(int bin, int pair, int NBT, int nquads, int chans, int c) {
  const int blk = pair / 64, in_blk = pair % 64, held = std::min(64, NBT - 64 * blk);
  return ((((size_t)blk * 64 * nquads + (size_t)(bin / 4) * held + in_blk) * chans + c) * 4 + (bin & 3)) * 2;
}

using namespace os2d_dft;

static std::vector<double> table(int n) {
  std::vector<double> t(2 * n);
  for (int m = 0; m < n; ++m) {
    t[2 * m] = std::cos(-2.0 * M_PI * m / n);
    t[2 * m + 1] = std::sin(-2.0 * M_PI * m / n);
  }
  return t;
}

static double frand(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xffffff) / double(1 << 24);
}

static int os2d_ws(int W) { return W + 3; }
static int os2d_base(int W) { return dft_round_up(3 * os2d_ws(W) + 3, 4); }
static int os2d_plane(int H, int W) { return dft_round_up(os2d_base(W) + (H + 3) * os2d_ws(W) + 3, 64); }

static int check_case(int H, int W, int C, int NB, int grid) {
  DftPlan pl;
  if (!dft_make_plan(H, W, &pl)) {
    std::printf("no plan for %dx%d\n", H, W);
    return 1;
  }
  const int P = pl.P, Q = pl.Q, V = pl.V, T = pl.T, NBT = NB * T, Cpad = dft_round_up(C, 8) + 0, HW = H * W;
  std::printf("case %dx%d C=%d NB=%d: P=%d Q=%d bins=%d tiles=%dx%d (%dx%d) window %dx%d lds=%d fast=%d\n", H, W, C, NB, P, Q, pl.NBINS,
              pl.TY, pl.TX, pl.TH, pl.TW, pl.LH, pl.LW, pl.lds_total, pl.fast);
  const std::vector<double> tp = table(P), tq = table(Q);
  std::vector<u32x4v> mats(dft_matrices_units(P, Q));
  const int nf = dft_units_fqt(P, Q), n2 = dft_units_fp2(P, Q), ne = dft_units_e2(P, Q), ng = dft_units_gq(P, Q);
  for (int i = 0; i < nf; ++i) dft_matrix_unit(0, i, P, Q, tp.data(), tq.data(), &mats[i]);
  for (int i = 0; i < n2; ++i) dft_matrix_unit(1, i, P, Q, tp.data(), tq.data(), &mats[nf + i]);
  for (int i = 0; i < ne; ++i) dft_matrix_unit(2, i, P, Q, tp.data(), tq.data(), &mats[nf + n2 + i]);
  for (int i = 0; i < ng; ++i) dft_matrix_unit(3, i, P, Q, tp.data(), tq.data(), &mats[nf + n2 + ne + i]);
  const u32x4v *FqT = mats.data(), *Fp2 = FqT + nf, *E2 = Fp2 + n2, *Gq = E2 + ne;

  // ---------------- forward
  unsigned seed = 12345u + H * 131 + W;
  std::vector<float> corr((size_t)NB * C * HW), invn((size_t)NB * HW);
  for (auto& v : corr) v = (float)(frand(seed) * 2.0 - 0.7);
  for (auto& v : invn) v = (float)(0.2 + 0.5 * frand(seed));          // relu(corr) * invn <= 1.3 * 0.7 < 1
  std::vector<float> X((size_t)(pl.NBINS / 4) * NBT * Cpad * 8, 777.0f);
  // both shapes of the kernels: 4 images per iteration (row operand of step 2 / step A in registers), and - where the transform takes
  // it - 8 images (row operand in LDS, whole activation units)
  DftPlan pl8;
  const bool have8 = dft_plan_g8(pl, &pl8) && Cpad % 8 == 0;
  if (!have8) std::printf("  (no 8-image plan for this transform)\n");
  for (int G = 4; G <= 8; G += 4) {
    if (G == 8 && !have8) continue;
    DftPlan fp = G == 8 ? pl8 : pl;
    const int CGf = (C + G - 1) / G, itf = NBT * CGf;
    fp.inv_cg = dft_magic((unsigned)CGf);
    std::fill(X.begin(), X.end(), 777.0f);
    emu::launch(grid, 512, fp.lds_total, [&] {
      if (G == 4) {
        // the k-step count of step 2 as a template parameter where the device build has one (dft_mfma.hip: dft_forward_pick)
#define FWD4(KS)                                                                                                             \
  if (fp.T > 1) dft_forward_body<true, false, 4, 8, KS>(corr.data(), invn.data(), X.data(), FqT, Fp2, fp, C, Cpad, NBT, itf);  \
  else if (fp.fast) dft_forward_body<false, true, 4, 8, KS>(corr.data(), invn.data(), X.data(), FqT, Fp2, fp, C, Cpad, NBT, itf); \
  else dft_forward_body<false, false, 4, 8, KS>(corr.data(), invn.data(), X.data(), FqT, Fp2, fp, C, Cpad, NBT, itf);
        switch (2 * fp.Pp / 16) {
          case 5: FWD4(5) break;
          case 6: FWD4(6) break;
          case 7: FWD4(7) break;
          case 8: FWD4(8) break;
          default: FWD4(0) break;
        }
#undef FWD4
      } else {
        if (fp.T > 1) dft_forward_body<true, false, 8, 8>(corr.data(), invn.data(), X.data(), FqT, Fp2, fp, C, Cpad, NBT, itf);
        else if (fp.fast) dft_forward_body<false, true, 8, 8>(corr.data(), invn.data(), X.data(), FqT, Fp2, fp, C, Cpad, NBT, itf);
        else dft_forward_body<false, false, 8, 8>(corr.data(), invn.data(), X.data(), FqT, Fp2, fp, C, Cpad, NBT, itf);
      }
    });
  double worst = 0.0, scale = 0.0;
  for (int nb = 0; nb < NB; ++nb)
    for (int tile = 0; tile < T; ++tile) {
      const int ty = tile / pl.TX, tx = tile % pl.TX;
      const int Y0 = T > 1 ? ty * pl.TH - pl.oy : 0, X0 = T > 1 ? tx * pl.TW - pl.ox : 0;
      for (int c = 0; c < C; ++c) {
        std::vector<double> x((size_t)pl.LH * pl.LW, 0.0);
        for (int r = 0; r < pl.LH; ++r)
          for (int cc = 0; cc < pl.LW; ++cc) {
            const int y = Y0 + r, xx = X0 + cc;
            if (y >= 0 && y < H && xx >= 0 && xx < W) {
              const float a = corr[((size_t)nb * C + c) * HW + y * W + xx], n = invn[(size_t)nb * HW + y * W + xx];
              x[(size_t)r * pl.LW + cc] = (double)(std::fmax(a, 0.f) * n);
            }
          }
        // row transforms then column transforms in float64
        std::vector<double> rr((size_t)pl.LH * V), ri((size_t)pl.LH * V);
        for (int r = 0; r < pl.LH; ++r)
          for (int v = 0; v < V; ++v) {
            double sr = 0, si = 0;
            for (int cc = 0; cc < pl.LW; ++cc) {
              const int a = (int)(((long long)v * cc) % Q);
              sr += x[(size_t)r * pl.LW + cc] * tq[2 * a];
              si += x[(size_t)r * pl.LW + cc] * tq[2 * a + 1];
            }
            rr[(size_t)r * V + v] = sr;
            ri[(size_t)r * V + v] = si;
          }
        const int pair = nb * T + tile;
        for (int u = 0; u < P; ++u)
          for (int v = 0; v < V; ++v) {
            double sr = 0, si = 0;
            for (int r = 0; r < pl.LH; ++r) {
              const int a = (int)(((long long)u * r) % P);
              const double cr = tp[2 * a], ci = tp[2 * a + 1];
              sr += rr[(size_t)r * V + v] * cr - ri[(size_t)r * V + v] * ci;
              si += rr[(size_t)r * V + v] * ci + ri[(size_t)r * V + v] * cr;
            }
            const int bin = v * P + u;
            const float* got = &X[spec_index(bin, pair, NBT, pl.NBINS / 4, Cpad, c)];
            worst = std::fmax(worst, std::fmax(std::fabs(got[0] - sr), std::fabs(got[1] - si)));
            scale = std::fmax(scale, std::fmax(std::fabs(sr), std::fabs(si)));
          }
        for (int bin = P * V; bin < pl.NBINS; ++bin) {
          const float* got = &X[spec_index(bin, pair, NBT, pl.NBINS / 4, Cpad, c)];
          if (got[0] != 0.f || got[1] != 0.f) {
            std::printf("padding bin %d not zero\n", bin);
            return 1;
          }
        }
      }
    }
  std::printf("  forward (G = %d): max |X - float64| = %.3e (largest |X| %.1f)\n", G, worst, scale);
  if (!(worst <= 2e-6 * scale + 1e-5)) {
    std::printf("FORWARD MISMATCH\n");
    return 1;
  }
  }

  // ---------------- inverse (Cout = 16 output channels: 4 / 2 groups)
  const int Cout = 16, MTP = 128, PLANE = os2d_plane(H, W), Ws = os2d_ws(W), BASE = os2d_base(W);
  std::vector<float> Y((size_t)(pl.NBINS / 4) * NBT * Cout * 8);
  for (size_t i = 0; i < Y.size(); ++i) {
    const double mag = std::exp(6.0 * frand(seed) - 2.0);              // a wide dynamic range between bins
    Y[i] = (float)((frand(seed) * 2.0 - 1.0) * mag);
  }
  for (int o = 0; o < Cout; ++o)                                       // and between images: scales 1e-3 .. 1e4
    for (size_t q = 0; q < (size_t)(pl.NBINS / 4) * NBT; ++q)
      for (int e = 0; e < 8; ++e) Y[(q * Cout + o) * 8 + e] *= (float)std::pow(10.0, (o % 8) - 3.0);
  // float64 inverse first: the channel scales of the epilogue are chosen from it the way the range plan does for the real
  // network (activations a few binades below 2^15: both fp16 halves normal)
  std::vector<double> yref((size_t)NB * Cout * HW, 0.0), ymaxo(Cout, 0.0);
  for (int nb = 0; nb < NB; ++nb)
    for (int o = 0; o < Cout; ++o)
      for (int tile = 0; tile < T; ++tile) {
        const int ty = tile / pl.TX, tx = tile % pl.TX, pair = nb * T + tile;
        const int y0 = T > 1 ? ty * pl.TH : 0, x0 = T > 1 ? tx * pl.TW : 0;
        const int TH_ = T > 1 ? pl.TH : H, TW_ = T > 1 ? pl.TW : W;
        for (int th = 0; th < TH_ && y0 + th < H; ++th) {
          const int hwin = th + pl.oy;
          std::vector<double> tr(V), ti(V);
          for (int v = 0; v < V; ++v) {
            double sr = 0, si = 0;
            for (int u = 0; u < P; ++u) {
              const int bin = v * P + u;
              const float* y = &Y[spec_index(bin, pair, NBT, pl.NBINS / 4, Cout, o)];
              const int a = (int)(((long long)u * hwin) % P);
              const double cr = tp[2 * a], ci = -tp[2 * a + 1];      // e^{+i}
              sr += y[0] * cr - y[1] * ci;
              si += y[0] * ci + y[1] * cr;
            }
            tr[v] = sr;
            ti[v] = si;
          }
          for (int tw = 0; tw < TW_ && x0 + tw < W; ++tw) {
            const int wwin = tw + pl.ox;
            double s2 = 0;
            for (int v = 0; v < V; ++v) {
              const double a_v = (v == 0 || 2 * v == Q) ? 1.0 : 2.0;
              const int a = (int)(((long long)v * wwin) % Q);
              s2 += a_v * (tr[v] * tq[2 * a] + ti[v] * tq[2 * a + 1]);     // Re(T e^{+i theta}) = Tr cos - Ti sin; tq = (cos, -sin)
            }
            const double yv = s2 / ((double)P * Q);
            yref[((size_t)nb * Cout + o) * HW + (size_t)(y0 + th) * W + x0 + tw] = yv;
            ymaxo[o] = std::fmax(ymaxo[o], std::fabs(yv));
          }
        }
      }
  std::vector<float> bp(3 * MTP, 0.f);
  for (int o = 0; o < Cout; ++o) {
    bp[o] = (float)(0.1 * ((o % 8) - 3) * ymaxo[o]);
    bp[2 * MTP + o] = (float)std::ldexp(1.0, (int)std::floor(std::log2(4096.0 / (1.4 * ymaxo[o]))));
  }
  for (int G = 4; G <= 8; G += 4) {
  if (G == 8 && !have8) continue;
  DftPlan ip = G == 8 ? pl8 : pl;
  std::vector<unsigned char> out((size_t)NB * ((Cout + 7) / 8) * 2 * PLANE * 16, 0x5A);     // a pattern: the kernel writes the borders too
  int flag = 0;
  const int OG = Cout / G, iters_i = NBT * OG;
  ip.inv_og = dft_magic((unsigned)OG);
  emu::launch(grid, DFT_THR, ip.lds_total, [&] {
#define INV(KS)                                                                                                                     \
  if (ip.T > 1) dft_inverse_body<true, KS>(Y.data(), bp.data(), MTP, out.data(), E2, Gq, ip, Cout, NBT, PLANE, Ws, BASE, iters_i, &flag, 1); \
  else dft_inverse_body<false, KS>(Y.data(), bp.data(), MTP, out.data(), E2, Gq, ip, Cout, NBT, PLANE, Ws, BASE, iters_i, &flag, 1);
    if (G == 8) {
      if (ip.T > 1) dft_inverse_body<true, 0, 8>(Y.data(), bp.data(), MTP, out.data(), E2, Gq, ip, Cout, NBT, PLANE, Ws, BASE, iters_i, &flag, 1);
      else dft_inverse_body<false, 0, 8>(Y.data(), bp.data(), MTP, out.data(), E2, Gq, ip, Cout, NBT, PLANE, Ws, BASE, iters_i, &flag, 1);
    } else
    switch (2 * ip.Pp / 16) {      // as dft_inverse_pick of the device build
      case 5: INV(5) break;
      case 6: INV(6) break;
      case 7: INV(7) break;
      case 8: INV(8) break;
      default: INV(0) break;
    }
#undef INV
  });
  double worst_rel = 0.0;
  for (int nb = 0; nb < NB; ++nb)
    for (int o = 0; o < Cout; ++o) {
      const unsigned char* hi = &out[(((size_t)nb * ((Cout + 7) / 8) + (o >> 3)) * 2 + 0) * (size_t)PLANE * 16];
      const unsigned char* lo = &out[(((size_t)nb * ((Cout + 7) / 8) + (o >> 3)) * 2 + 1) * (size_t)PLANE * 16];
      const double unit = ymaxo[o] * (double)bp[2 * MTP + o];
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
          const size_t cell = (size_t)BASE + (size_t)h * Ws + w;
          const _Float16 hv = *reinterpret_cast<const _Float16*>(hi + cell * 16 + (o & 7) * 2);
          const _Float16 lv = *reinterpret_cast<const _Float16*>(lo + cell * 16 + (o & 7) * 2);
          const double want = std::fmax(yref[((size_t)nb * Cout + o) * HW + (size_t)h * W + w] + (double)bp[o], 0.0) * (double)bp[2 * MTP + o];
          worst_rel = std::fmax(worst_rel, std::fabs((double)hv + (double)lv - want) / unit);
        }
    }
  // the kernel also writes the zero borders of the planes (the buffer was filled with a pattern)
  for (int nb = 0; nb < NB; ++nb)
    for (int pln = 0; pln < ((Cout + 7) / 8) * 2; ++pln) {
      const unsigned char* base = &out[((size_t)nb * ((Cout + 7) / 8) * 2 + pln) * (size_t)PLANE * 16];
      for (int cell = 0; cell < PLANE; ++cell) {
        const int rel = cell - BASE, hh = rel >= 0 ? rel / Ws : -1, ww = rel >= 0 ? rel % Ws : -1;
        if (rel >= 0 && hh < H && ww < W) continue;
        for (int b = 0; b < 16; ++b)
          if (base[(size_t)cell * 16 + b] != 0) {
            std::printf("border cell %d of plane %d (pair %d) not zero\n", cell, pln, nb);
            return 1;
          }
      }
    }
  std::printf("  inverse (G = %d): max |y - float64| / max |y| = %.3e, flag %d\n", G, worst_rel, flag);
  if (!(worst_rel <= 1.5e-6) || flag != 0) {
    std::printf("INVERSE MISMATCH\n");
    return 1;
  }
  }
  return 0;
}

int main(int argc, char** argv) {
  int rc = 0;
  if (argc >= 2 && argv[1][0] == 'c') {      // "canonical" as the first argument: plan on the six canonical transform sizes
    emu_dft_policy = 1;
    --argc;
    ++argv;
  }
  if (argc >= 5) {
    for (int i = 1; i + 3 < argc; i += 4) rc |= check_case(std::atoi(argv[i]), std::atoi(argv[i + 1]), std::atoi(argv[i + 2]), std::atoi(argv[i + 3]), 2);
  } else {
    rc |= check_case(11, 13, 5, 1, 2);       // small, W % 4 != 0, a partial channel group
    rc |= check_case(20, 24, 4, 2, 3);       // fast path, two pairs, more work-groups than one iteration each
  }
  std::printf(rc ? "FAILED\n" : "ok\n");
  return rc;
}

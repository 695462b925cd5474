/* oracle/nufft_oracle_impl.h -- precision-generic body of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY (see nufft_oracle.c header). Included twice by
 * nufft_oracle.c with
 *     FLT  = float  / double          working real type
 *     SUF(name) = name##_f32 / name##_f64
 * Every function cites the reference lines it restates
 * (paths relative to /root/reference/tensorflow_nufft/cc/kernels/).
 */

typedef struct { FLT re, im; } SUF(cplx);
#define CPLX SUF(cplx)

/* ---------------------------------------------------------------- kernel */

/* ES kernel by the defining formula. Reference: nufft_util.cc:64-69
 * (evaluate_kernel) and nufft_plan.cc:1243-1289 (evaluate_kernel_vector,
 * kerevalmeth 0). Evaluated in the working precision, like the reference. */
static inline FLT SUF(es_kernel)(FLT x, const kernel_params *kp) {
  if (FABS(x) >= (FLT)kp->half_width) return (FLT)0;
  return EXP((FLT)kp->beta * SQRT((FLT)1 - (FLT)kp->c * x * x));
}

/* Fill ker[0..w-1] = phi(x1 + j). x1 in [-w/2, -w/2+1].
 * method 0: formula (reference kerevalmeth 0, nufft_plan.cc:1243-1289).
 * method 1: piecewise polynomial in z = 2*x1 + w - 1 evaluated by Horner's
 *           rule (reference eval_kernel_vec_Horner, nufft_plan.cc:1291-1307),
 *           but with coefficients fitted here (fit_horner_table) instead of
 *           the reference's generated .inc tables. */
static inline void SUF(eval_kernel_vec)(FLT *ker, FLT x1,
                                        const kernel_params *kp) {
  const int w = kp->w;
  if (kp->method == 0) {
    for (int j = 0; j < w; ++j) ker[j] = SUF(es_kernel)(x1 + (FLT)j, kp);
  } else {
    const FLT z = (FLT)2 * x1 + (FLT)w - (FLT)1;
    const int nc = kp->ncoef;
    FLT acc[ORACLE_MAX_W];
    const FLT *tab = kp->SUF(horner);          /* table in the working precision */
    for (int j = 0; j < ORACLE_MAX_W; ++j) acc[j] = tab[(nc - 1) * ORACLE_MAX_W + j];
    for (int k = nc - 2; k >= 0; --k)          /* j loop vectorises (reference pads to 4) */
      for (int j = 0; j < ORACLE_MAX_W; ++j)
        acc[j] = acc[j] * z + tab[k * ORACLE_MAX_W + j];
    for (int j = 0; j < w; ++j) ker[j] = acc[j];
  }
}

/* ------------------------------------------------------------- fold/scale */

/* Reference: FoldAndRescale functors, nufft_plan.h:676-734, and the CPU
 * macro of the same meaning used inside the sort/spread loops
 * (nufft_plan.cc:497-499, 1087-1089). Result in [0, n]. */
static inline FLT SUF(fold_rescale)(FLT x, int64_t n, int range) {
  const FLT pi = (FLT)M_PI;
  FLT s;
  if (range == ORACLE_RANGE_STRICT) {
    s = x + pi;
  } else if (range == ORACLE_RANGE_EXTENDED) {
    if (x > pi) s = x - pi;
    else if (x < -pi) s = x + (FLT)3 * pi;
    else s = x + pi;
  } else {
    s = FMOD(x + pi, (FLT)2 * pi);
    if (s < (FLT)0) s += (FLT)2 * pi;
  }
  return s * (FLT)(1.0 / (2.0 * M_PI)) * (FLT)n;
}

/* --------------------------------------------------------------- bin sort */

/* Counting sort of points into bins of 16 x 4 x 4 fine cells, x fastest.
 * Output perm[sorted position] = original index. Stable.
 * Reference: binsort_singlethread nufft_plan.cc:475-531; the multi-threaded
 * variant (:533-652) yields the same permutation, so the threads below only
 * split the counting/filling work the same way (per-thread chunk counts). */
static void SUF(binsort)(int64_t M, const FLT *const xyz[3], int rank,
                         const int64_t nf[3], int range, int32_t *perm,
                         int nthreads) {
  const int bin_dims[3] = {16, 4, 4};
  int64_t box[3] = {1, 1, 1};
  int64_t nbins = 1;
  for (int d = 0; d < rank; ++d) {
    box[d] = nf[d] / bin_dims[d] + 1; /* +1: round-off near +pi (:484-486) */
    nbins *= box[d];
  }
  if (nthreads < 1) nthreads = 1;
  if ((int64_t)nthreads > M) nthreads = (int)(M > 0 ? M : 1);
  int64_t *counts = (int64_t *)calloc((size_t)nbins * nthreads, sizeof(int64_t));
  int64_t *brk = (int64_t *)malloc(sizeof(int64_t) * (nthreads + 1));
  for (int t = 0; t <= nthreads; ++t)
    brk[t] = (int64_t)(0.5 + (double)M * t / (double)nthreads);
  int32_t *binof = (int32_t *)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));

#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
  for (int t = 0; t < nthreads; ++t) {
    int64_t *ct = counts + (size_t)t * nbins;
    for (int64_t i = brk[t]; i < brk[t + 1]; ++i) {
      int64_t b = 0;
      for (int d = rank - 1; d >= 0; --d) {
        int64_t bi = (int64_t)(SUF(fold_rescale)(xyz[d][i], nf[d], range) /
                               (FLT)bin_dims[d]);
        b = bi + box[d] * b;
      }
      binof[i] = (int32_t)b;
      ct[b]++;
    }
  }
  /* offsets: bin-major, thread-minor => stable */
  int64_t run = 0;
  for (int64_t b = 0; b < nbins; ++b)
    for (int t = 0; t < nthreads; ++t) {
      int64_t c = counts[(size_t)t * nbins + b];
      counts[(size_t)t * nbins + b] = run;
      run += c;
    }
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
  for (int t = 0; t < nthreads; ++t) {
    int64_t *ct = counts + (size_t)t * nbins;
    for (int64_t i = brk[t]; i < brk[t + 1]; ++i)
      perm[ct[binof[i]]++] = (int32_t)i;
  }
  free(binof);
  free(brk);
  free(counts);
}

/* ---------------------------------------------------------------- spread */

/* Bounding box of a subproblem's points, padded for the kernel.
 * Reference: get_subgrid nufft_plan.cc:1736-1804. */
static void SUF(get_subgrid)(int64_t off[3], int64_t size[3], int64_t M,
                             FLT *const k[3], int w, int rank) {
  const FLT ns2 = (FLT)w / 2;
  for (int d = 0; d < 3; ++d) { off[d] = 0; size[d] = 1; }
  for (int d = 0; d < rank; ++d) {
    FLT lo = k[d][0], hi = k[d][0];
    for (int64_t i = 1; i < M; ++i) {
      if (k[d][i] < lo) lo = k[d][i];
      if (k[d][i] > hi) hi = k[d][i];
    }
    off[d] = (int64_t)CEIL(lo - ns2);
    size[d] = (int64_t)CEIL(hi - ns2) - off[d] + w;
  }
}

/* Spread one subproblem into its private, unwrapped subgrid du.
 * Reference: spread_subproblem_1d/2d/3d nufft_plan.cc:1463-1636 (stencil
 * start i = ceil(x - w/2), offset x1 = i - x clipped into [-w/2, -w/2+1],
 * rank-1 update of interleaved (re,im) rows). */
static void SUF(spread_subproblem)(const int64_t off[3], const int64_t size[3],
                                   FLT *du, int64_t M, FLT *const k[3],
                                   const FLT *dd, int rank,
                                   const kernel_params *kp) {
  const int w = kp->w;
  const FLT ns2 = (FLT)w / 2;
  const int64_t ntot = size[0] * size[1] * size[2];
  for (int64_t i = 0; i < 2 * ntot; ++i) du[i] = 0;
  FLT ker[3][ORACLE_MAX_W];
  FLT ker1val[2 * ORACLE_MAX_W];
  for (int64_t i = 0; i < M; ++i) {
    const FLT re0 = dd[2 * i], im0 = dd[2 * i + 1];
    int64_t i0[3] = {0, 0, 0};
    for (int d = 0; d < rank; ++d) {
      i0[d] = (int64_t)CEIL(k[d][i] - ns2);
      FLT x1 = (FLT)i0[d] - k[d][i];
      if (x1 < -ns2) x1 = -ns2;          /* :1499-1500 */
      if (x1 > -ns2 + 1) x1 = -ns2 + 1;
      SUF(eval_kernel_vec)(ker[d], x1, kp);
    }
    for (int j = 0; j < w; ++j) {
      ker1val[2 * j] = re0 * ker[0][j];
      ker1val[2 * j + 1] = im0 * ker[0][j];
    }
    if (rank == 1) {
      FLT *t = du + 2 * (i0[0] - off[0]);
      for (int l = 0; l < 2 * w; ++l) t[l] += ker1val[l];
    } else if (rank == 2) {
      for (int dy = 0; dy < w; ++dy) {
        FLT *t = du + 2 * (size[0] * (i0[1] - off[1] + dy) + i0[0] - off[0]);
        const FLT kv = ker[1][dy];
        for (int l = 0; l < 2 * w; ++l) t[l] += kv * ker1val[l];
      }
    } else {
      for (int dz = 0; dz < w; ++dz) {
        const int64_t oz = size[0] * size[1] * (i0[2] - off[2] + dz);
        for (int dy = 0; dy < w; ++dy) {
          FLT *t = du + 2 * (oz + size[0] * (i0[1] - off[1] + dy) + i0[0] - off[0]);
          const FLT kv = ker[1][dy] * ker[2][dz];
          for (int l = 0; l < 2 * w; ++l) t[l] += kv * ker1val[l];
        }
      }
    }
  }
}

/* Add a subgrid into the periodic fine grid.
 * Reference: add_wrapped_subgrid nufft_plan.cc:1638-1682. */
static void SUF(add_wrapped_subgrid)(const int64_t off[3],
                                     const int64_t size[3],
                                     const int64_t nf[3], FLT *fw,
                                     const FLT *du) {
  int64_t *o[3];
  for (int d = 0; d < 3; ++d) {
    o[d] = (int64_t *)malloc(sizeof(int64_t) * (size_t)size[d]);
    for (int64_t i = 0; i < size[d]; ++i) {
      int64_t x = off[d] + i;
      x %= nf[d];
      if (x < 0) x += nf[d];
      o[d][i] = x;
    }
  }
  for (int64_t dz = 0; dz < size[2]; ++dz)
    for (int64_t dy = 0; dy < size[1]; ++dy) {
      const int64_t ro = nf[0] * (o[1][dy] + nf[1] * o[2][dz]);
      const FLT *s = du + 2 * size[0] * (dy + size[1] * dz);
      for (int64_t dx = 0; dx < size[0]; ++dx) {
        const int64_t j = ro + o[0][dx];
        fw[2 * j] += s[2 * dx];
        fw[2 * j + 1] += s[2 * dx + 1];
      }
    }
  for (int d = 0; d < 3; ++d) free(o[d]);
}

/* The same with one atomic update per element, for concurrent callers.
 * Reference: add_wrapped_subgrid_thread_safe nufft_plan.cc:1685-1734, chosen
 * over the critical section when more than atomic_threshold = 10 threads
 * spread (nufft_plan.cc:923, :1109-1114). */
static void SUF(add_wrapped_subgrid_atomic)(const int64_t off[3],
                                            const int64_t size[3],
                                            const int64_t nf[3], FLT *fw,
                                            const FLT *du) {
  int64_t *o[3];
  for (int d = 0; d < 3; ++d) {
    o[d] = (int64_t *)malloc(sizeof(int64_t) * (size_t)size[d]);
    for (int64_t i = 0; i < size[d]; ++i) {
      int64_t x = (off[d] + i) % nf[d];
      if (x < 0) x += nf[d];
      o[d][i] = x;
    }
  }
  for (int64_t dz = 0; dz < size[2]; ++dz)
    for (int64_t dy = 0; dy < size[1]; ++dy) {
      const int64_t ro = nf[0] * (o[1][dy] + nf[1] * o[2][dz]);
      const FLT *s = du + 2 * size[0] * (dy + size[1] * dz);
      for (int64_t dx = 0; dx < size[0]; ++dx) {
        const int64_t j = ro + o[0][dx];
#pragma omp atomic
        fw[2 * j] += s[2 * dx];
#pragma omp atomic
        fw[2 * j + 1] += s[2 * dx + 1];
      }
    }
  for (int d = 0; d < 3; ++d) free(o[d]);
}

/* Type-1 spreading of sorted points onto the (zeroed) fine grid.
 * Reference: spreadSorted nufft_plan.cc:1027-1132 (subproblem split :1053-1071,
 * gather by permutation :1085-1092, merge :1109-1114, spread-only scale
 * :1126-1129). */
static void SUF(spread_sorted)(const int32_t *perm, const int64_t nf[3],
                               FLT *fw, int64_t M, const FLT *const xyz[3],
                               const FLT *c, int rank, int range,
                               const kernel_params *kp, int nthreads,
                               double scale) {
  const int64_t N = nf[0] * nf[1] * nf[2];
  for (int64_t i = 0; i < 2 * N; ++i) fw[i] = 0;
  if (M == 0) return;
  if (nthreads < 1) nthreads = 1;
  const int64_t max_sub = (rank == 1) ? 10000 : 100000; /* :919 */
  int64_t nb = nthreads < M ? nthreads : M;
  if (nb * max_sub < M) nb = 1 + (M - 1) / max_sub;
  if (M * 1000 < N) nb = M; /* low-density rescue :1062-1065 */
  int64_t *brk = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nb + 1));
  for (int64_t p = 0; p <= nb; ++p)
    brk[p] = (int64_t)(0.5 + (double)M * (double)p / (double)nb);

#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
  for (int64_t isub = 0; isub < nb; ++isub) {
    const int64_t M0 = brk[isub + 1] - brk[isub];
    if (M0 <= 0) continue;
    FLT *k0[3] = {NULL, NULL, NULL};
    for (int d = 0; d < rank; ++d)
      k0[d] = (FLT *)malloc(sizeof(FLT) * (size_t)M0);
    FLT *dd0 = (FLT *)malloc(sizeof(FLT) * 2 * (size_t)M0);
    for (int64_t j = 0; j < M0; ++j) {
      const int64_t kk = perm[j + brk[isub]];
      for (int d = 0; d < rank; ++d)
        k0[d][j] = SUF(fold_rescale)(xyz[d][kk], nf[d], range);
      dd0[2 * j] = c[2 * kk];
      dd0[2 * j + 1] = c[2 * kk + 1];
    }
    int64_t off[3], size[3];
    SUF(get_subgrid)(off, size, M0, k0, kp->w, rank);
    FLT *du0 = (FLT *)malloc(sizeof(FLT) * 2 * (size_t)(size[0] * size[1] * size[2]));
    SUF(spread_subproblem)(off, size, du0, M0, k0, dd0, rank, kp);
    if (nthreads > 10) { /* atomic_threshold, nufft_plan.cc:923, :1109-1114 */
      SUF(add_wrapped_subgrid_atomic)(off, size, nf, fw, du0);
    } else {
#pragma omp critical(oracle_add_wrapped)
      SUF(add_wrapped_subgrid)(off, size, nf, fw, du0);
    }
    free(du0);
    free(dd0);
    for (int d = 0; d < rank; ++d) free(k0[d]);
  }
  free(brk);
  if (scale != 1.0)
    for (int64_t i = 0; i < 2 * N; ++i) fw[i] *= (FLT)scale;
}

/* ---------------------------------------------------------------- interp */

/* Type-2 interpolation from the fine grid at the sorted points.
 * Reference: interpSorted nufft_plan.cc:1136-1240 with interp_line/square/
 * cube :1309-1461 (same stencil rule as spreading, periodic wrap of indices,
 * interp-only scale :1223-1226). One general wrapped path is used here. */
static void SUF(interp_sorted)(const int32_t *perm, const int64_t nf[3],
                               const FLT *fw, int64_t M,
                               const FLT *const xyz[3], FLT *c, int rank,
                               int range, const kernel_params *kp,
                               int nthreads, double scale) {
  const int w = kp->w;
  const FLT ns2 = (FLT)w / 2;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1000)
  for (int64_t i = 0; i < M; ++i) {
    const int64_t j = perm[i];
    FLT ker[3][ORACLE_MAX_W];
    int64_t idx[3][ORACLE_MAX_W];
    for (int d = 0; d < 3; ++d) { ker[d][0] = 1; idx[d][0] = 0; }
    for (int d = 0; d < rank; ++d) {
      const FLT xj = SUF(fold_rescale)(xyz[d][j], nf[d], range);
      const int64_t i0 = (int64_t)CEIL(xj - ns2);
      FLT x1 = (FLT)i0 - xj;
      if (x1 < -ns2) x1 = -ns2;
      if (x1 > -ns2 + 1) x1 = -ns2 + 1;
      SUF(eval_kernel_vec)(ker[d], x1, kp);
      for (int l = 0; l < w; ++l) {
        int64_t x = (i0 + l) % nf[d];
        if (x < 0) x += nf[d];
        idx[d][l] = x;
      }
    }
    const int wy = rank > 1 ? w : 1, wz = rank > 2 ? w : 1;
    FLT re = 0, im = 0;
    for (int dz = 0; dz < wz; ++dz)
      for (int dy = 0; dy < wy; ++dy) {
        const int64_t ro = nf[0] * (idx[1][dy] + nf[1] * idx[2][dz]);
        const FLT kyz = ker[1][dy] * ker[2][dz];
        FLT lre = 0, lim = 0;
        for (int dx = 0; dx < w; ++dx) {
          const int64_t g = ro + idx[0][dx];
          lre += fw[2 * g] * ker[0][dx];
          lim += fw[2 * g + 1] * ker[0][dx];
        }
        re += kyz * lre;
        im += kyz * lim;
      }
    c[2 * j] = re * (FLT)scale;
    c[2 * j + 1] = im * (FLT)scale;
  }
}

/* ------------------------------------------------------------------- FFT */

/* Unnormalised in-place complex DFT with exponent sign `sign`, applied along
 * every axis of an x-fastest array. Stands in for the FFTW plan the reference
 * builds at nufft_plan.cc:363-430 (fftw plan_many_dft, in place, sign =
 * fft_direction) and executes at :336. Mixed radix 2/3/4/5 (+ generic),
 * decimation in time; twiddles always computed in double. */
static void SUF(fft_rec)(int64_t n, int64_t is, const CPLX *in, CPLX *out,
                         const CPLX *tw, int64_t tws) {
  if (n == 1) { out[0] = in[0]; return; }
  int r;
  if (n % 4 == 0) r = 4;
  else if (n % 2 == 0) r = 2;
  else if (n % 3 == 0) r = 3;
  else if (n % 5 == 0) r = 5;
  else { r = 7; while (n % r) r += 2; }
  const int64_t m = n / r;
  for (int q = 0; q < r; ++q)
    SUF(fft_rec)(m, is * r, in + q * is, out + q * m, tw, tws * r);
  /* butterflies: X[k + p m] = sum_q W_n^{qk} W_r^{pq} Y_q[k] */
  CPLX y[16], *yy = y;
  CPLX *heap = NULL;
  if (r > 16) { heap = (CPLX *)malloc(sizeof(CPLX) * (size_t)r); yy = heap; }
  for (int64_t k = 0; k < m; ++k) {
    for (int q = 0; q < r; ++q) {
      const CPLX t = tw[(q * k) * tws];
      const CPLX v = out[q * m + k];
      yy[q].re = v.re * t.re - v.im * t.im;
      yy[q].im = v.re * t.im + v.im * t.re;
    }
    for (int p = 0; p < r; ++p) {
      FLT sre = yy[0].re, sim = yy[0].im;
      for (int q = 1; q < r; ++q) {
        /* W_r^{pq} = tw[((p*q) % r) * m * tws] */
        const CPLX t = tw[(((int64_t)p * q) % r) * m * tws];
        sre += yy[q].re * t.re - yy[q].im * t.im;
        sim += yy[q].re * t.im + yy[q].im * t.re;
      }
      out[p * m + k].re = sre;
      out[p * m + k].im = sim;
    }
  }
  if (heap) free(heap);
}

static void SUF(fft_nd)(CPLX *a, const int64_t nf[3], int rank, int sign,
                        int nthreads) {
  if (nthreads < 1) nthreads = 1;
  int64_t stride = 1;
  const int64_t total = nf[0] * nf[1] * nf[2];
  for (int d = 0; d < rank; ++d) {
    const int64_t n = nf[d];
    CPLX *tw = (CPLX *)malloc(sizeof(CPLX) * (size_t)n);
    for (int64_t j = 0; j < n; ++j) {
      const double ang = (double)sign * 2.0 * M_PI * (double)j / (double)n;
      tw[j].re = (FLT)cos(ang);
      tw[j].im = (FLT)sin(ang);
    }
    const int64_t nlines = total / n;
#pragma omp parallel num_threads(nthreads)
    {
      CPLX *bi = (CPLX *)malloc(sizeof(CPLX) * (size_t)n);
      CPLX *bo = (CPLX *)malloc(sizeof(CPLX) * (size_t)n);
#pragma omp for schedule(static)
      for (int64_t l = 0; l < nlines; ++l) {
        /* line l: inner index (below axis d) and outer index (above) */
        const int64_t inner = l % stride, outer = l / stride;
        CPLX *base = a + inner + outer * stride * n;
        for (int64_t j = 0; j < n; ++j) bi[j] = base[j * stride];
        SUF(fft_rec)(n, 1, bi, bo, tw, 1);
        for (int64_t j = 0; j < n; ++j) base[j * stride] = bo[j];
      }
      free(bi);
      free(bo);
    }
    free(tw);
    stride *= n;
  }
}

/* ------------------------------------------------------------ deconvolve */

/* Type 1 (dir 1): f[k] = fw[k mod nf] / prod_d phihat_d[|k_d|], CMCL order
 * (array index 0 <-> most negative mode). Type 2 (dir 2): zero fw then the
 * inverse assignment. Reference: deconvolve_1d/2d/3d nufft_plan.cc:729-881. */
static void SUF(deconvolve)(int dir, CPLX *f, CPLX *fw, const int64_t N[3],
                            const int64_t nf[3], int rank,
                            double *const fser[3]) {
  const int64_t nftot = nf[0] * nf[1] * nf[2];
  if (dir == 2) memset(fw, 0, sizeof(CPLX) * (size_t)nftot);
  int64_t kmin[3] = {0, 0, 0};
  for (int d = 0; d < rank; ++d) kmin[d] = -(N[d] / 2);
  for (int64_t a2 = 0; a2 < N[2]; ++a2) {
    const int64_t k2 = rank > 2 ? kmin[2] + a2 : 0;
    const int64_t w2 = k2 >= 0 ? k2 : nf[2] + k2;
    const double s2 = rank > 2 ? fser[2][k2 < 0 ? -k2 : k2] : 1.0;
    for (int64_t a1 = 0; a1 < N[1]; ++a1) {
      const int64_t k1 = rank > 1 ? kmin[1] + a1 : 0;
      const int64_t w1 = k1 >= 0 ? k1 : nf[1] + k1;
      const double s1 = rank > 1 ? fser[1][k1 < 0 ? -k1 : k1] : 1.0;
      for (int64_t a0 = 0; a0 < N[0]; ++a0) {
        const int64_t k0 = kmin[0] + a0;
        const int64_t w0 = k0 >= 0 ? k0 : nf[0] + k0;
        const FLT s = (FLT)(fser[0][k0 < 0 ? -k0 : k0] * s1 * s2);
        const int64_t fi = a0 + N[0] * (a1 + N[1] * a2);
        const int64_t wi = w0 + nf[0] * (w1 + nf[1] * w2);
        if (dir == 1) {
          f[fi].re = fw[wi].re / s;
          f[fi].im = fw[wi].im / s;
        } else {
          fw[wi].re = f[fi].re / s;
          fw[wi].im = f[fi].im / s;
        }
      }
    }
  }
}

/* ------------------------------------------------------------- transform */

/* Whole transform: plan + set_points + execute of the reference CPU path
 * (Plan<CPU>::initialize nufft_plan.cc:166-265, set_points :267-302,
 * execute :316-351), one transform at a time. */
int SUF(oracle_nufft)(const oracle_opts *o, int64_t M, const FLT *x,
                      const FLT *y, const FLT *z, FLT *c, FLT *f,
                      oracle_info *info) {
  kernel_params kp;
  int64_t N[3] = {1, 1, 1}, nf[3] = {1, 1, 1};
  int rc = oracle_setup(o, (int)sizeof(FLT), &kp, N, nf, info);
  if (rc) return rc;
  const int rank = o->rank;
  const FLT *xyz[3] = {x, y, z};
  const int nthreads = o->nthreads > 0 ? o->nthreads : omp_get_max_threads();
  const int64_t Ntot = N[0] * N[1] * N[2], nftot = nf[0] * nf[1] * nf[2];

  double *fser[3] = {NULL, NULL, NULL};
  for (int d = 0; d < rank; ++d) {
    fser[d] = (double *)malloc(sizeof(double) * (size_t)(nf[d] / 2 + 1));
    oracle_kernel_fseries(nf[d], &kp, fser[d]);
  }
  /* ORACLE_TIMING=1: stage times of a type-1 call on stderr (bench tooling) */
  const int timing = getenv("ORACLE_TIMING") != NULL;
  double tm[5];
  tm[0] = omp_get_wtime();
  int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));
  SUF(binsort)(M, xyz, rank, nf, o->points_range, perm, nthreads);
  CPLX *fw = (CPLX *)malloc(sizeof(CPLX) * (size_t)nftot);
  tm[1] = omp_get_wtime();

  for (int t = 0; t < o->ntransf; ++t) {
    FLT *ct = c + 2 * (size_t)t * (size_t)M;
    CPLX *ft = (CPLX *)f + (size_t)t * (size_t)Ntot;
    if (o->type == 1) {
      SUF(spread_sorted)(perm, nf, (FLT *)fw, M, xyz, ct, rank,
                         o->points_range, &kp, nthreads, 1.0);
      tm[2] = omp_get_wtime();
      SUF(fft_nd)(fw, nf, rank, o->iflag, nthreads);
      tm[3] = omp_get_wtime();
      SUF(deconvolve)(1, ft, fw, N, nf, rank, fser);
      tm[4] = omp_get_wtime();
      if (timing)
        fprintf(stderr, "oracle type 1, %d threads: sort %.1f ms, spread %.1f, fft %.1f, deconvolve %.1f\n", nthreads,
                1e3 * (tm[1] - tm[0]), 1e3 * (tm[2] - tm[1]), 1e3 * (tm[3] - tm[2]), 1e3 * (tm[4] - tm[3]));
    } else {
      SUF(deconvolve)(2, ft, fw, N, nf, rank, fser);
      SUF(fft_nd)(fw, nf, rank, o->iflag, nthreads);
      SUF(interp_sorted)(perm, nf, (const FLT *)fw, M, xyz, ct, rank,
                         o->points_range, &kp, nthreads, 1.0);
    }
  }
  free(fw);
  free(perm);
  for (int d = 0; d < rank; ++d) free(fser[d]);
  return 0;
}

/* Spread-only / interp-only ops (no upsampling, no FFT, scaled by
 * calculate_scale_factor). Reference: Plan<CPU>::spread / interp
 * nufft_plan.cc:353-361 region and the scale at :1126-1129, :1223-1226. */
int SUF(oracle_spread_interp)(const oracle_opts *o, int64_t M, const FLT *x,
                              const FLT *y, const FLT *z, FLT *c, FLT *f,
                              oracle_info *info) {
  kernel_params kp;
  int64_t N[3] = {1, 1, 1}, nf[3] = {1, 1, 1};
  oracle_opts oo = *o;
  oo.spread_only = 1;
  int rc = oracle_setup(&oo, (int)sizeof(FLT), &kp, N, nf, info);
  if (rc) return rc;
  const int rank = o->rank;
  const FLT *xyz[3] = {x, y, z};
  const int nthreads = o->nthreads > 0 ? o->nthreads : omp_get_max_threads();
  const int64_t nftot = nf[0] * nf[1] * nf[2];
  const double scale = oracle_scale_factor(rank, &kp);
  int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));
  SUF(binsort)(M, xyz, rank, nf, o->points_range, perm, nthreads);
  for (int t = 0; t < o->ntransf; ++t) {
    FLT *ct = c + 2 * (size_t)t * (size_t)M;
    FLT *ft = f + 2 * (size_t)t * (size_t)nftot;
    if (o->type == 1)
      SUF(spread_sorted)(perm, nf, ft, M, xyz, ct, rank, o->points_range, &kp,
                         nthreads, scale);
    else
      SUF(interp_sorted)(perm, nf, ft, M, xyz, ct, rank, o->points_range, &kp,
                         nthreads, scale);
  }
  free(perm);
  return 0;
}

/* Stage entry points used by bench.py's cpu_baseline leg and by tests. */
void SUF(oracle_binsort)(int64_t M, const FLT *x, const FLT *y, const FLT *z,
                         int rank, const int64_t *nf, int range, int32_t *perm,
                         int nthreads) {
  const FLT *xyz[3] = {x, y, z};
  int64_t n3[3] = {1, 1, 1};
  for (int d = 0; d < rank; ++d) n3[d] = nf[d];
  SUF(binsort)(M, xyz, rank, n3, range, perm,
               nthreads > 0 ? nthreads : omp_get_max_threads());
}

int SUF(oracle_spread_stage)(const oracle_opts *o, const int32_t *perm,
                             int64_t M, const FLT *x, const FLT *y,
                             const FLT *z, const FLT *c, FLT *fw) {
  kernel_params kp;
  int64_t N[3], nf[3];
  oracle_info info;
  int rc = oracle_setup(o, (int)sizeof(FLT), &kp, N, nf, &info);
  if (rc) return rc;
  const FLT *xyz[3] = {x, y, z};
  SUF(spread_sorted)(perm, nf, fw, M, xyz, c, o->rank, o->points_range, &kp,
                     o->nthreads > 0 ? o->nthreads : omp_get_max_threads(),
                     1.0);
  return 0;
}

void SUF(oracle_fft)(FLT *a, const int64_t *nf, int rank, int sign,
                     int nthreads) {
  int64_t n3[3] = {1, 1, 1};
  for (int d = 0; d < rank; ++d) n3[d] = nf[d];
  SUF(fft_nd)((CPLX *)a, n3, rank, sign,
              nthreads > 0 ? nthreads : omp_get_max_threads());
}

void SUF(oracle_eval_kernel)(const oracle_opts *o, int n, const FLT *x1,
                             FLT *out) {
  kernel_params kp;
  int64_t N[3], nf[3];
  oracle_info info;
  if (oracle_setup(o, (int)sizeof(FLT), &kp, N, nf, &info)) return;
  for (int i = 0; i < n; ++i)
    SUF(eval_kernel_vec)(out + (size_t)i * kp.w, x1[i], &kp);
}

#undef CPLX

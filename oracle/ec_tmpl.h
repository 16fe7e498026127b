/*
 * ec_tmpl.h — short-Weierstrass (a = 0) group law over a field given by the F_*
 * macros; included twice by bn254_oracle.c (G1 over Fq, G2 over Fq2).
 * TEST INFRASTRUCTURE ONLY (see bn254_oracle.c header).
 *
 * Restates icicle/include/icicle/curves/projective.h (complete Renes–Costello–
 * Batina formulas in homogeneous projective coordinates, identity = (0,1,0)) and
 * icicle/include/icicle/curves/affine.h (affine identity = (0,0)).
 */

typedef struct { F_T x, y; } PFX(aff);
typedef struct { F_T x, y, z; } PFX(proj);

/* Projective::zero() — projective.h:25 */
static inline void PFX(set_zero)(PFX(proj)* r)
{
  F_SETZERO(&r->x);
  F_SETONE(&r->y);
  F_SETZERO(&r->z);
}
static inline int PFX(aff_is_zero)(const PFX(aff)* a) { return F_ISZERO(&a->x) && F_ISZERO(&a->y); }
/* Projective::from_affine — projective.h:34-37 */
static inline void PFX(from_affine)(PFX(proj)* r, const PFX(aff)* a)
{
  if (PFX(aff_is_zero)(a)) { PFX(set_zero)(r); return; }
  r->x = a->x;
  r->y = a->y;
  F_SETONE(&r->z);
}
/* Projective::to_affine — projective.h:27-31 (inverse(0)=0 ⇒ identity ↦ (0,0)) */
static inline void PFX(to_affine)(PFX(aff)* r, const PFX(proj)* p)
{
  F_T d;
  F_INV(&d, &p->z);
  F_MUL(&r->x, &p->x, &d);
  F_MUL(&r->y, &p->y, &d);
}
/* Projective::neg — projective.h:52 */
static inline void PFX(neg)(PFX(proj)* r, const PFX(proj)* p)
{
  r->x = p->x;
  F_NEG(&r->y, &p->y);
  r->z = p->z;
}
static inline void PFX(aff_neg)(PFX(aff)* r, const PFX(aff)* p)
{
  r->x = p->x;
  F_NEG(&r->y, &p->y);
}
/* Projective::is_zero — projective.h:220-223 */
static inline int PFX(is_zero)(const PFX(proj)* p) { return F_ISZERO(&p->x) && !F_ISZERO(&p->y) && F_ISZERO(&p->z); }

/* Projective::dbl — projective.h:54-80 */
static void PFX(dbl)(PFX(proj)* r, const PFX(proj)* p)
{
  F_T X = p->x, Y = p->y, Z = p->z, t0, t1, t2, X3, Y3, Z3;
  F_MUL(&t0, &Y, &Y);
  F_ADD(&Z3, &t0, &t0);
  F_ADD(&Z3, &Z3, &Z3);
  F_ADD(&Z3, &Z3, &Z3);
  F_MUL(&t1, &Y, &Z);
  F_MUL(&t2, &Z, &Z);
  F_MULB3(&t2, &t2);
  F_MUL(&X3, &t2, &Z3);
  F_ADD(&Y3, &t0, &t2);
  F_MUL(&Z3, &t1, &Z3);
  F_ADD(&t1, &t2, &t2);
  F_ADD(&t2, &t1, &t2);
  F_SUB(&t0, &t0, &t2);
  F_MUL(&Y3, &t0, &Y3);
  F_ADD(&Y3, &X3, &Y3);
  F_MUL(&t1, &X, &Y);
  F_MUL(&X3, &t0, &t1);
  F_ADD(&X3, &X3, &X3);
  r->x = X3; r->y = Y3; r->z = Z3;
}

/* Projective + Projective — projective.h:82-128 */
static void PFX(add)(PFX(proj)* r, const PFX(proj)* p1, const PFX(proj)* p2)
{
  F_T X1 = p1->x, Y1 = p1->y, Z1 = p1->z, X2 = p2->x, Y2 = p2->y, Z2 = p2->z;
  F_T t00, t01, t02, t03, t04, t05, t06, t07, t08, t09, t10, t11, t12, t13, t14, t15, t16, t17, t18, t19, t20, t21, t22, t23;
  F_T a, b, X3, Y3, Z3;
  F_MUL(&t00, &X1, &X2);
  F_MUL(&t01, &Y1, &Y2);
  F_MUL(&t02, &Z1, &Z2);
  F_ADD(&t03, &X1, &Y1);
  F_ADD(&t04, &X2, &Y2);
  F_MUL(&t05, &t03, &t04);
  F_ADD(&t06, &t00, &t01);
  F_SUB(&t07, &t05, &t06);
  F_ADD(&t08, &Y1, &Z1);
  F_ADD(&t09, &Y2, &Z2);
  F_MUL(&t10, &t08, &t09);
  F_ADD(&t11, &t01, &t02);
  F_SUB(&t12, &t10, &t11);
  F_ADD(&t13, &X1, &Z1);
  F_ADD(&t14, &X2, &Z2);
  F_MUL(&t15, &t13, &t14);
  F_ADD(&t16, &t00, &t02);
  F_SUB(&t17, &t15, &t16);
  F_ADD(&t18, &t00, &t00);
  F_ADD(&t19, &t18, &t00);
  F_MULB3(&t20, &t02);
  F_ADD(&t21, &t01, &t20);
  F_SUB(&t22, &t01, &t20);
  F_MULB3(&t23, &t17);
  F_MUL(&a, &t12, &t23);
  F_MUL(&b, &t07, &t22);
  F_SUB(&X3, &b, &a);
  F_MUL(&a, &t23, &t19);
  F_MUL(&b, &t22, &t21);
  F_ADD(&Y3, &b, &a);
  F_MUL(&a, &t19, &t07);
  F_MUL(&b, &t21, &t12);
  F_ADD(&Z3, &b, &a);
  r->x = X3; r->y = Y3; r->z = Z3;
}

/* Projective + Affine — projective.h:132-169.  NOTE: exactly as the reference, this
 * formula treats the affine operand as (x, y, 1); callers skip the affine identity
 * (0,0) before calling (cpu_msm.hpp:276, cuda_msm.cuh:251). */
static void PFX(add_mixed)(PFX(proj)* r, const PFX(proj)* p1, const PFX(aff)* p2)
{
  F_T X1 = p1->x, Y1 = p1->y, Z1 = p1->z, X2 = p2->x, Y2 = p2->y, one;
  F_T t00, t01, t02, t03, t04, t05, t06, t07, t08, t09, t10, t11, t12, t13, t14, t15, t16, t17, t18, t19, t20, t21, t22, t23;
  F_T a, b, X3, Y3, Z3;
  F_SETONE(&one);
  F_MUL(&t00, &X1, &X2);
  F_MUL(&t01, &Y1, &Y2);
  t02 = Z1;
  F_ADD(&t03, &X1, &Y1);
  F_ADD(&t04, &X2, &Y2);
  F_MUL(&t05, &t03, &t04);
  F_ADD(&t06, &t00, &t01);
  F_SUB(&t07, &t05, &t06);
  F_ADD(&t08, &Y1, &Z1);
  F_ADD(&t09, &Y2, &one);
  F_MUL(&t10, &t08, &t09);
  F_ADD(&t11, &t01, &t02);
  F_SUB(&t12, &t10, &t11);
  F_ADD(&t13, &X1, &Z1);
  F_ADD(&t14, &X2, &one);
  F_MUL(&t15, &t13, &t14);
  F_ADD(&t16, &t00, &t02);
  F_SUB(&t17, &t15, &t16);
  F_ADD(&t18, &t00, &t00);
  F_ADD(&t19, &t18, &t00);
  F_MULB3(&t20, &t02);
  F_ADD(&t21, &t01, &t20);
  F_SUB(&t22, &t01, &t20);
  F_MULB3(&t23, &t17);
  F_MUL(&a, &t12, &t23);
  F_MUL(&b, &t07, &t22);
  F_SUB(&X3, &b, &a);
  F_MUL(&a, &t23, &t19);
  F_MUL(&b, &t22, &t21);
  F_ADD(&Y3, &b, &a);
  F_MUL(&a, &t19, &t07);
  F_MUL(&b, &t21, &t12);
  F_ADD(&Z3, &b, &a);
  r->x = X3; r->y = Y3; r->z = Z3;
}

/* operator== — projective.h:210-213 */
static int PFX(eq)(const PFX(proj)* p1, const PFX(proj)* p2)
{
  F_T a, b;
  F_MUL(&a, &p1->x, &p2->z);
  F_MUL(&b, &p2->x, &p1->z);
  if (!F_EQ(&a, &b)) return 0;
  F_MUL(&a, &p1->y, &p2->z);
  F_MUL(&b, &p2->y, &p1->z);
  return F_EQ(&a, &b);
}

/* is_on_curve — projective.h:225-232 (b = b3/3 is not stored; check 3·(Z·Y² − X³) = b3·Z³) */
static int PFX(is_on_curve)(const PFX(proj)* p)
{
  if (PFX(is_zero)(p)) return 1;
  if (F_ISZERO(&p->z)) return 0;
  F_T z3, x3, zy2, lhs, rhs, t;
  F_MUL(&t, &p->z, &p->z);
  F_MUL(&z3, &t, &p->z);
  F_MUL(&t, &p->x, &p->x);
  F_MUL(&x3, &t, &p->x);
  F_MUL(&t, &p->y, &p->y);
  F_MUL(&zy2, &t, &p->z);
  F_SUB(&t, &zy2, &x3);
  F_ADD(&lhs, &t, &t);
  F_ADD(&lhs, &lhs, &t);
  F_MULB3(&rhs, &z3);
  return F_EQ(&lhs, &rhs);
}

/* scalar · point, fixed 4-bit windows — projective.h:176-208 */
static void PFX(mul_scalar)(PFX(proj)* r, const PFX(proj)* point, const u256* scalar)
{
  PFX(proj) table[15], res;
  table[0] = *point;
  for (int i = 1; i < 15; i++) PFX(add)(&table[i], &table[i - 1], point);
  PFX(set_zero)(&res);
  int nz = 0;
  for (int w = 63; w >= 0; w--) { /* nof_windows = ceil(254/4) = 64 */
    uint32_t d = u256_digit(scalar, (unsigned)w, 4);
    for (int j = 0; nz && j < 4; j++) PFX(dbl)(&res, &res);
    if (d) {
      PFX(add)(&res, &res, &table[d - 1]);
      nz = 1;
    }
  }
  *r = res;
}

/* Σ sᵢ·Pᵢ by definition */
static void PFX(msm_naive)(PFX(proj)* out, const u256* scalars, const PFX(aff)* bases, int n)
{
  PFX(proj) acc, p, t;
  PFX(set_zero)(&acc);
  for (int i = 0; i < n; i++) {
    PFX(from_affine)(&p, &bases[i]);
    PFX(mul_scalar)(&t, &p, &scalars[i]);
    PFX(add)(&acc, &acc, &t);
  }
  *out = acc;
}

/* Pippenger bucket method — icicle/backend/cpu/src/curve/cpu_msm.hpp:41-430.
 *  phase 1 (:258-330)  signed c-bit digits with carry; a scalar whose top bit is set is
 *                      replaced by r − s and its base negated (:282-283); affine-zero bases
 *                      are skipped (:276); every worker owns a private bucket array over a
 *                      contiguous slice of the input (:240-247).
 *  phase 2 (:333-…)    buckets of each window ("bucket module") are collapsed with the
 *                      running-sum ("line sum / triangle sum") trick.
 *  phase 3             window sums are combined by Horner's rule with c doublings. */
static void PFX(msm)(PFX(proj)* out, const u256* scalars, const PFX(aff)* bases, int n, int c)
{
  if (c <= 0) {
    int lg = 0;
    while ((1 << lg) < n) lg++;
    c = lg - 3;
    if (c < 1) c = 1;
    if (c > 16) c = 16;
  }
  const int nbm = (254 - 1) / c + 1 + 1; /* +1: room for the final carry */
  const int bm_size = 1 << (c - 1);      /* buckets per window: digit magnitudes 1..2^(c-1) */
  const size_t nb = (size_t)nbm * bm_size;
  int T = 1;
#ifdef _OPENMP
  T = omp_get_max_threads();
#endif
  /* worker layout: the reference gives every worker a private bucket array over a slice of the input
   * (cpu_msm.hpp:240-247).  With many cores that costs T full bucket arrays; here the T workers are a
   * (windows × slices) grid instead — worker (w, sl) owns window w of slice sl — so memory and the merge
   * step scale with the number of slices, not with T. */
  int slices = T / nbm;
  if (slices < 1) slices = 1;
  if (slices > n) slices = n > 0 ? n : 1;
  PFX(proj)* buckets = (PFX(proj)*)malloc(sizeof(PFX(proj)) * nb * slices);
  uint8_t* busy = (uint8_t*)calloc(nb * slices, 1);
  const int per = (n + slices - 1) / slices;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int sl = 0; sl < slices; sl++) {
    for (int w = 0; w < nbm; w++) {
      PFX(proj)* B = buckets + (size_t)sl * nb + (size_t)w * bm_size;
      uint8_t* busyB = busy + (size_t)sl * nb + (size_t)w * bm_size;
      int lo = sl * per, hi = lo + per < n ? lo + per : n;
      for (int i = lo; i < hi; i++) {
        if (PFX(aff_is_zero)(&bases[i])) continue;
        u256 s = scalars[i];
        int negate = (int)((s.l[3] >> 61) & 1); /* bit 253 = top bit of a 254-bit scalar */
        if (negate) fp_neg(&FR, &s, &s);
        /* signed digit of window w: d_w = raw_w + carry_w, carry chain recomputed from window 0 */
        uint32_t carry = 0, d = 0;
        for (int v = 0; v <= w; v++) {
          d = u256_digit(&s, (unsigned)v, (unsigned)c) + carry;
          carry = d > (uint32_t)bm_size ? 1 : 0;
        }
        if (d == 0) continue;
        int neg_digit = 0;
        if (d > (uint32_t)bm_size) { /* digit in (2^(c-1), 2^c] → d − 2^c, carry 1 */
          d = (1u << c) - d;
          neg_digit = 1;
          if (d == 0) continue; /* d was exactly 2^c */
        }
        PFX(aff) base_neg;
        const PFX(aff)* P = &bases[i];
        if (negate ^ neg_digit) { PFX(aff_neg)(&base_neg, &bases[i]); P = &base_neg; }
        if (busyB[d - 1]) PFX(add_mixed)(&B[d - 1], &B[d - 1], P);
        else { PFX(from_affine)(&B[d - 1], P); busyB[d - 1] = 1; }
      }
    }
  }
  /* merge slice copies into copy 0 */
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < (int64_t)nb; k++) {
    for (int t = 1; t < slices; t++) {
      size_t o = (size_t)t * nb + k;
      if (!busy[o]) continue;
      if (busy[k]) PFX(add)(&buckets[k], &buckets[k], &buckets[o]);
      else { buckets[k] = buckets[o]; busy[k] = 1; }
    }
  }
  /* per-window running sums: Σ_d d·B[d] */
  PFX(proj)* wsum = (PFX(proj)*)malloc(sizeof(PFX(proj)) * nbm);
#pragma omp parallel for schedule(dynamic, 1)
  for (int w = 0; w < nbm; w++) {
    PFX(proj) line, tri;
    PFX(set_zero)(&line);
    PFX(set_zero)(&tri);
    for (int d = bm_size; d >= 1; d--) {
      size_t idx = (size_t)w * bm_size + (d - 1);
      if (busy[idx]) PFX(add)(&line, &line, &buckets[idx]);
      PFX(add)(&tri, &tri, &line);
    }
    wsum[w] = tri;
  }
  /* Horner */
  PFX(proj) res;
  PFX(set_zero)(&res);
  for (int w = nbm - 1; w >= 0; w--) {
    for (int j = 0; j < c; j++) PFX(dbl)(&res, &res);
    PFX(add)(&res, &res, &wsum[w]);
  }
  *out = res;
  free(wsum);
  free(busy);
  free(buckets);
}

/* out[i] = s[i]·base in affine form (8-bit fixed windows over a precomputed table) */
static void PFX(fixed_base_mul)(PFX(aff)* out, const PFX(aff)* base, const u256* s, int64_t n)
{
  PFX(proj)* table = (PFX(proj)*)malloc(sizeof(PFX(proj)) * 32 * 255);
  PFX(proj) cur;
  PFX(from_affine)(&cur, base);
  for (int w = 0; w < 32; w++) {
    PFX(proj)* row = table + (size_t)w * 255;
    row[0] = cur;
    for (int d = 1; d < 255; d++) PFX(add)(&row[d], &row[d - 1], &cur);
    PFX(add)(&cur, &row[254], &cur); /* 256·cur */
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    PFX(proj) acc;
    PFX(set_zero)(&acc);
    for (int w = 0; w < 32; w++) {
      uint32_t d = u256_digit(&s[i], (unsigned)w, 8);
      if (d) PFX(add)(&acc, &acc, &table[(size_t)w * 255 + d - 1]);
    }
    PFX(to_affine)(&out[i], &acc);
  }
  free(table);
}

/* oracle/orc_internal.h -- helpers shared between the oracle's translation units.
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#define ORC_SVD_MAXN 12
#define ORC_SVD_MAXM 12

void orc_svd_backsubst_vec(int m, int n, const double *w, const double *Ut, const double *Vt,
                           const double *b, double *x);
void orc_svd_solve(const double *A, int m, int n, const double *b, double *x);
void orc_svd_invert(const double *A, int n, double *Ainv);

#endif

/* Plain-C restatement of the reference's prune/regrow update, full-sort form.
 *
 * TEST INFRASTRUCTURE ONLY (second, independent pin of oracle/rigl_oracle.py
 * and the "two O(n log n) sorts per layer" cost model of the CPU baseline).
 * Follows rigl/sparse_optimizers_base.py:276-343 line by line:
 *   :286-290  n_ones, n_prune = (int32)((float)n_ones * drop_fraction), n_keep
 *   :293-302  top_k(score_drop, k = n) -> mask1 = first n_keep sorted indices
 *   :307-310  lifted = mask1 ? min(score_grow) - 1 : score_grow
 *   :311-318  top_k(lifted, k = n)     -> mask2 = first n_prune sorted indices
 *   :320-321  assert sum(mask1 * mask2) == 0
 *   :328-335  new = mask2 & (reinit ? 1 : mask == 0); w[new] = grow value
 *   :345-353 / :555-564  momentum[new] = reset value
 *   :340-342  mask = mask1 + mask2
 * tf.nn.top_k order: value descending, equal values by LOWER index first.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const float* g_key;

static int cmp_desc_stable(const void* a, const void* b) {
  const int32_t ia = *(const int32_t*)a, ib = *(const int32_t*)b;
  const float va = g_key[ia], vb = g_key[ib];
  if (vb < va) return -1;
  if (vb > va) return 1;
  return (ia > ib) - (ia < ib);
}

/* counts: [0] n_ones [1] n_prune [2] n_keep [3] n_new [4] overlap */
int rigl_oracle_update(int64_t n, const float* score_drop, const float* score_grow, const float* mask, float* w,
                       float* momentum /* nullable */, const float* grow_values /* nullable: zeros */,
                       const float* momentum_values /* nullable: zeros */, float drop_fraction, int reinit_when_same,
                       float* new_mask, int32_t* counts) {
  if (n <= 0) return 0;
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  float* lifted = (float*)malloc(sizeof(float) * (size_t)n);
  unsigned char* m1 = (unsigned char*)calloc((size_t)n, 1);
  unsigned char* m2 = (unsigned char*)calloc((size_t)n, 1);
  if (!idx || !lifted || !m1 || !m2) return -1;
  float sum = 0.f;                    /* fp32 reduce_sum; exact for n < 2^24 */
  for (int64_t i = 0; i < n; ++i) sum += mask[i];
  const int32_t n_ones = (int32_t)sum;
  const int32_t n_prune = (int32_t)((float)n_ones * drop_fraction);
  const int32_t n_keep = n_ones - n_prune;
  for (int64_t i = 0; i < n; ++i) idx[i] = (int32_t)i;
  g_key = score_drop;
  qsort(idx, (size_t)n, sizeof(int32_t), cmp_desc_stable);
  for (int32_t i = 0; i < n_keep && i < n; ++i) m1[idx[i]] = 1;
  float gmin = score_grow[0];
  for (int64_t i = 1; i < n; ++i) if (score_grow[i] < gmin) gmin = score_grow[i];
  const float low = gmin - 1.0f;
  for (int64_t i = 0; i < n; ++i) lifted[i] = m1[i] ? low : score_grow[i];
  for (int64_t i = 0; i < n; ++i) idx[i] = (int32_t)i;
  g_key = lifted;
  qsort(idx, (size_t)n, sizeof(int32_t), cmp_desc_stable);
  for (int32_t i = 0; i < n_prune && i < n; ++i) m2[idx[i]] = 1;
  int32_t overlap = 0, n_new = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (m1[i] && m2[i]) overlap = 1;
    const int is_new = m2[i] && (reinit_when_same || mask[i] == 0.f);
    if (is_new) {
      ++n_new;
      w[i] = grow_values ? grow_values[i] : 0.f;
      if (momentum) momentum[i] = momentum_values ? momentum_values[i] : 0.f;
    }
    new_mask[i] = (float)(m1[i] + m2[i]);
  }
  counts[0] = n_ones; counts[1] = n_prune; counts[2] = n_keep; counts[3] = n_new; counts[4] = overlap;
  free(idx); free(lifted); free(m1); free(m2);
  return 0;
}

/* rumi_voc.h — C ABI of the MI355X bag-of-words transform.
 *
 * Replaces, for the hot call `mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)` of Frame::ComputeBoW /
 * KeyFrame::ComputeBoW (R/lib_src/Frame.cc:763-768, KeyFrame.cc:245-252):
 *   DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features, BowVector&, FeatureVector&, levelsup)
 *     R/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1126-1199, the per-feature tree descent :1217-1260,
 *   DBoW2::FORB::distance               R/Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101,
 *   DBoW2::BowVector::addWeight / addIfNotExist / normalize   R/Thirdparty/DBoW2/DBoW2/BowVector.cpp:34-84,
 *   DBoW2::FeatureVector::addFeature    R/Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:31-45.
 * The tree descent (n features x L levels x k children, 256-bit Hamming arg-min) runs on the GPU; the two ordered maps are
 * assembled on the host in feature order, exactly as the reference's loop does (sums in feature order, L1/L2 norm in word order).
 *
 * Status codes as in rumi_orb.h.  No CPU fallback. */
#ifndef RUMI_VOC_H
#define RUMI_VOC_H
#include <stdint.h>

#include "rumi_orb.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RumiVocabulary RumiVocabulary;

/* Build a vocabulary from the fields of the reference's text format (TemplatedVocabulary::loadFromTextFile,
 * TemplatedVocabulary.h:1338-1425): node 0 is the root; for node i >= 1: parent[i], is_leaf[i], desc[i][32], weight[i], in file
 * order (children are visited in increasing node id, word ids are given to leaves in file order).  weighting / scoring are the
 * header's n2 / n1 (DBoW2::WeightingType TF_IDF=0, TF=1, IDF=2, BINARY=3; ScoringType L1_NORM=0 .. DOT_PRODUCT=5). */
int rumi_voc_create(int32_t n_nodes, const int32_t *parent, const uint8_t *is_leaf, const uint8_t *desc, const double *weight,
                    int32_t weighting, int32_t scoring, int32_t device, RumiVocabulary **out);
/* The same from an ORBvoc.txt-style file (first line "k L scoring weighting"); m_L is the header's L, as in DBoW2, not the tree's depth. */
int rumi_voc_load_text(const char *path, int32_t device, RumiVocabulary **out);
void rumi_voc_destroy(RumiVocabulary *v);
int32_t rumi_voc_words(const RumiVocabulary *v);
int32_t rumi_voc_levels(const RumiVocabulary *v); /* m_L: depth of the deepest leaf (rumi_voc_create) or the file header's L (rumi_voc_load_text) */
int rumi_voc_set_levels(RumiVocabulary *v, int32_t L); /* override m_L (1..10), e.g. for a tree built in memory from a file's nodes */

/* transform(feature, word_id, weight, &nid, levelsup) for n descriptors (host arrays).  node_id[i] = the node on the path at
 * level L - levelsup (0 = root when that level is <= 0; 0 as well where DBoW2 would leave it unset: a leaf above that level). */
int rumi_voc_transform_features(RumiVocabulary *v, const uint8_t *desc, int32_t n, int32_t levelsup, uint32_t *word_id, double *weight,
                                uint32_t *node_id);
/* The same for the device-resident output of rumi_orb_extract_batch_device: d_desc [nframes][cap][32], d_counts [nframes][2]
 * (n, monoIndex); results d_word / d_node [nframes][cap] (uint32), d_weight [nframes][cap] (double); entries >= n untouched. */
int rumi_voc_transform_batch_device(RumiVocabulary *v, const void *d_desc, const void *d_counts, int32_t nframes, int32_t cap,
                                    int32_t levelsup, void *d_word, void *d_weight, void *d_node, void *hip_stream);

/* The whole transform(features, BowVector&, FeatureVector&, levelsup): BowVector as (word id ascending, value) pairs, FeatureVector
 * in the CSR form rumi_search_by_bow takes (RumiFeatureVector of rumi_match.h: node ids ascending, offsets, feature indices).
 * Capacities: bow_* and fv_nodes [n], fv_offsets [n + 1], fv_indices [n]. */
int rumi_voc_transform(RumiVocabulary *v, const uint8_t *desc, int32_t n, int32_t levelsup, uint32_t *bow_ids, double *bow_vals,
                       int32_t *n_words_out, uint32_t *fv_nodes, int32_t *fv_offsets, uint32_t *fv_indices, int32_t *n_nodes_out);

/* The second half of rumi_voc_transform alone: BowVector and FeatureVector from the per-feature results (word_id, weight, node_id of
 * rumi_voc_transform_features / rumi_track_reference_keyframe), assembled in feature order exactly as TemplatedVocabulary::transform's loop
 * does (:1147-1190).  Host-only. */
int rumi_voc_assemble(const RumiVocabulary *v, int32_t n, const uint32_t *word_id, const double *weight, const uint32_t *node_id, uint32_t *bow_ids,
                      double *bow_vals, int32_t *n_words_out, uint32_t *fv_nodes, int32_t *fv_offsets, uint32_t *fv_indices, int32_t *n_nodes_out);

#ifdef __cplusplus
}
#endif
#endif

// placeholder until family 2 lands (replaced in the next commit)
#include "malcs_internal.hpp"
extern "C" {
int malcs_lcs_graph_batch(malcs_batch *, const int64_t *, const int64_t *, malcs_graphs **) { malcs_set_error("not implemented"); return MALCS_ERR_INTERNAL; }
int malcs_graphs_destroy(malcs_graphs *) { return MALCS_OK; }
int malcs_graphs_counts(const malcs_graphs *, int64_t *) { return MALCS_ERR_INTERNAL; }
int malcs_graphs_status(const malcs_graphs *, int32_t *, int64_t *) { return MALCS_ERR_INTERNAL; }
int malcs_graphs_supremal(const malcs_graphs *, int64_t *) { return MALCS_ERR_INTERNAL; }
int malcs_graphs_runs(const malcs_graphs *, uint64_t, int64_t *) { return MALCS_ERR_INTERNAL; }
int malcs_graphs_nodes(const malcs_graphs *, uint64_t, int64_t *) { return MALCS_ERR_INTERNAL; }
int malcs_graphs_edges(const malcs_graphs *, uint64_t, int64_t *) { return MALCS_ERR_INTERNAL; }
int malcs_graphs_timing(const malcs_graphs *, float *) { return MALCS_ERR_INTERNAL; }
}

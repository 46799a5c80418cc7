/*
 * rrx_rccl.h -- the one collective of the path below Python: all-gather of column-sharded flux arrays over RCCL (xGMI).
 *
 * The reference has no multi-GPU code (its host model MicroHH decomposes the domain itself and never
 * exchanges radiation data); columns are independent, so ranks own contiguous column ranges and the only exchange is the
 * gather of the broadband fluxes a caller may want on every rank (rrx_allgather_fluxes).
 * Library: rte-rrtmgp-cpp_amd/lib/librrx_rccl.so (links librccl; librrx_hip.so itself has no RCCL dependency).
 *
 * One process per GPU. Bootstrap: rank 0 calls rrx_comm_get_unique_id and hands the 128 bytes to the other ranks by any means
 * (the C++ driver uses a file, rrx_comm_id_to_file / rrx_comm_id_from_file); every rank then calls rrx_comm_create after
 * selecting its device. All functions return 0 on success; rrx_rccl_last_error() gives the message otherwise.
 */
#ifndef RRX_RCCL_H
#define RRX_RCCL_H
#ifdef __cplusplus
extern "C" {
#endif

#define RRX_COMM_ID_BYTES 128

const char* rrx_rccl_last_error(void);
int rrx_comm_get_unique_id(char id[RRX_COMM_ID_BYTES]);
/* atomically publish / wait for (timeout_s seconds) the id through a file on a filesystem all ranks see */
int rrx_comm_id_to_file(const char* path, const char id[RRX_COMM_ID_BYTES]);
int rrx_comm_id_from_file(const char* path, char id[RRX_COMM_ID_BYTES], int timeout_s);
int rrx_comm_create(int world, int rank, const char id[RRX_COMM_ID_BYTES], void** comm);
int rrx_comm_destroy(void* comm);

/* Contiguous column range [*col_s, *col_e) (0-based) of `rank` when ncol_total columns are split over `world` ranks: the first
   ncol_total % world ranks own one column more (same rule as rte-rrtmgp-cpp_amd/sharding.py:column_range). */
void rrx_column_range(int rank, int world, int ncol_total, int* col_s, int* col_e);

/* local: (nrows, ncol_local) with the column fastest, e.g. nrows = nflux*nlev of a packed (nflux, nlev, ncol_local) array;
   gathered: (nrows, ncol_total) on every rank; scratch: (world + 1) * nrows * ceil(ncol_total / world) words of device memory.
   One ncclAllGather of equal-sized (padded) blocks + one kernel that places each rank's columns. */
int rrx_allgather_fluxes_f64(void* comm, int nrows, int ncol_total, const double* local, double* gathered, double* scratch, void* stream);
int rrx_allgather_fluxes_f32(void* comm, int nrows, int ncol_total, const float* local, float* gathered, float* scratch, void* stream);

/* layout check of the pad / place kernels for `world` ranks on ONE device (no communicator): 0 = the gathered array is the original */
int rrx_rccl_selftest_layout(int world, int nrows, int ncol_total);

#ifdef __cplusplus
}
#endif
#endif

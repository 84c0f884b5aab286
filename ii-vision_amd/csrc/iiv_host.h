// iiv_host.h -- host-side declarations shared by the translation units of
// libiivision.so (the C ABI itself is include/iivision.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/iivision.h"
#include "iiv_device.h"

namespace iiv {

int set_error(int code, const char *fmt, ...);
int hip_check(hipError_t e, const char *what);

#define IIV_HIP(expr)                                    \
    do {                                                 \
        int _rc = ::iiv::hip_check((expr), #expr);       \
        if (_rc) return _rc;                             \
    } while (0)

// iiv_tables.hip
int cie2000_matrix(const uint8_t rgb[48], double out_f[256], int32_t out_i[256], hipStream_t st);
int symmetrise_table(int mode, uint16_t *d_table, hipStream_t st);
int store_table_from_table(int mode, const uint16_t *d_table, uint16_t *d_store, hipStream_t st);
int delta_e_pairs(int n, const double *lab1, const double *lab2, double *out, hipStream_t st);
int pixel_strings(int mode, uint32_t *d_dots, uint8_t *d_pixels, ulonglong2 *d_strings, hipStream_t st);
int build_table(int mode, const int32_t dm[256], uint16_t *d_out, int symmetric, hipStream_t st);
int build_store_table(int mode, const int32_t dm[256], uint16_t *d_out, hipStream_t st);
int build_strings(int mode, const int32_t dm[256], ulonglong2 **d_strings, uint16_t **d_sub, hipStream_t st);
int build_hgr_dot_lut(uint32_t **d_out, hipStream_t st);   // HGR: windows -> dots, two lookups (iiv_edit.h: hgr_dot_slot_lo)
int check_dw_piece_table(int mode, const int32_t dm[256], const uint16_t *d_table, unsigned long long *mismatches, hipStream_t st);
int build_dw_piece_table(int mode, const uint16_t *d_sub, uint32_t **d_out, hipStream_t st);   // DHGR: [2 banks][4096] u32; HGR: [4096]
size_t split_entries(int mode, int right);
int build_split_tables(int mode, const ulonglong2 *d_strings, const uint16_t *d_sub, uint32_t *d_left, uint32_t *d_right,
                       hipStream_t st);
struct NarrowTables;
int build_narrow_tables(int mode, const ulonglong2 *d_strings, const uint16_t *d_sub, const uint32_t *d_left, const uint16_t *d_store,
                        NarrowTables *out, hipStream_t st);
void free_narrow_tables(NarrowTables *nt);
int expand_narrow_tables(int mode, const NarrowTables &nt, const uint16_t *d_store, uint16_t *d_out, unsigned long long *n_mismatch,
                         hipStream_t st);
int build_narrow_store_table(int mode, const int32_t dm[256], const uint16_t *d_store, uint16_t *d_expanded,
                             unsigned long long *n_mismatch, hipStream_t st);
int transpose_split_tables(int mode, const uint32_t *d_left, const uint32_t *d_right, uint32_t *d_left_t,
                           uint32_t *d_right_t, hipStream_t st);
struct NarrowTables;
int build_joint_tables(int mode, const NarrowTables &nt, uint32_t **d_jl, uint32_t **d_jr, hipStream_t st);
size_t split_dw_entries(int mode, int right);
int build_split_dw_tables(int mode, const ulonglong2 *d_strings, const uint16_t *d_sub, uint32_t *d_left, uint32_t *d_right,
                          hipStream_t st);
int check_split_dw_table(int mode, const int32_t dm[256], const uint16_t *d_table, unsigned long long *mismatches,
                         hipStream_t st);
int build_split_store_table(int mode, const int32_t dm[256], uint32_t *d_left, uint32_t *d_right, uint16_t *d_expanded,
                            hipStream_t st);

// iiv_bitmap.hip
int pack(int mode, int n, const uint8_t *d_main, const uint8_t *d_aux, uint64_t *d_packed, hipStream_t st);
int diff_weights(int mode, const uint16_t *d_table, int n, const uint64_t *d_src, const uint64_t *d_tgt,
                 int is_aux, int32_t *d_out, hipStream_t st);
int compute_delta_pages(int mode, const uint16_t *d_table, int n, const uint64_t *d_tgt, const int32_t *d_pages,
                        const int32_t *d_contents, const int32_t *d_dw_rows, int is_aux, int32_t *d_out,
                        hipStream_t st);

}  // namespace iiv

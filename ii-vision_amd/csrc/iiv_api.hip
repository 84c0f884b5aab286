// iiv_api.hip -- extern "C" entry points of libiivision.so that are not tied to
// the encoder object (those live in iiv_encode.hip), plus error plumbing.
#include "iiv_host.h"

#include <stdarg.h>

namespace iiv {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_check(hipError_t e, const char *what)
{
    if (e == hipSuccess) return IIV_OK;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice)
        return set_error(IIV_ERR_NO_DEVICE, "%s: %s", what, hipGetErrorString(e));
    return set_error(IIV_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

static int check_mode(int mode)
{
    if (mode != kHGR && mode != kDHGR) return set_error(IIV_ERR_INVALID, "mode must be IIV_HGR or IIV_DHGR");
    return IIV_OK;
}

static int check_dm(int mode, const int32_t dm[256])
{
    if (!dm) return set_error(IIV_ERR_INVALID, "dm is NULL");
    int mx = 0;
    for (int i = 0; i < 256; i++) {
        if (dm[i] < 0) return set_error(IIV_ERR_INVALID, "dm[%d] < 0", i);
        mx = dm[i] > mx ? dm[i] : mx;
    }
    // every table value must fit the 11-bit fields the encoder packs it into
    if (mx * masked_dots(mode) > 2047)
        return set_error(IIV_ERR_INVALID, "max(dm) * MASKED_DOTS = %d exceeds 2047", mx * masked_dots(mode));
    return IIV_OK;
}

}  // namespace iiv

extern "C" {

#ifndef IIV_BUILD_ID
#define IIV_BUILD_ID "unknown"
#endif
// "iivision-gfx950 <version> build <id>": id = hash of the csrc/ sources, headers and compiler flags this library was built
// from (csrc/Makefile: BUILD_ID) -- what a committed counter run must carry to be quoted beside a bench line of this build
const char *iiv_version(void) { return "iivision-gfx950 0.2 build " IIV_BUILD_ID; }
const char *iiv_last_error(void) { return iiv::g_err; }

int iiv_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int iiv_masked_bits(int mode) { return iiv::masked_bits(mode); }
int iiv_masked_dots(int mode) { return iiv::masked_dots(mode); }
int iiv_num_offsets(int mode) { return iiv::num_offsets(mode); }
size_t iiv_table_entries(int mode) { return (size_t)iiv::num_offsets(mode) << (2 * iiv::masked_bits(mode)); }
size_t iiv_store_table_entries(int mode)
{
    return (size_t)iiv::num_offsets(mode) << (iiv::content_bits(mode) + iiv::masked_bits(mode));
}

int iiv_cie2000_matrix(const uint8_t rgb[48], double out_f[256], int32_t out_i[256], void *stream)
{
    if (!rgb) return iiv::set_error(IIV_ERR_INVALID, "rgb is NULL");
    return iiv::cie2000_matrix(rgb, out_f, out_i, (hipStream_t)stream);
}

int iiv_delta_e_cie2000(int n, const double *lab1, const double *lab2, double *out, void *stream)
{
    if (n < 0 || !lab1 || !lab2 || !out) return iiv::set_error(IIV_ERR_INVALID, "iiv_delta_e_cie2000: bad argument");
    if (n == 0) return IIV_OK;
    return iiv::delta_e_pairs(n, lab1, lab2, out, (hipStream_t)stream);
}

int iiv_pixel_strings(int mode, uint32_t *d_dots, uint8_t *d_pixels, void *stream)
{
    int rc = iiv::check_mode(mode);
    if (rc) return rc;
    return iiv::pixel_strings(mode, d_dots, d_pixels, nullptr, (hipStream_t)stream);
}

int iiv_build_table(int mode, const int32_t dm[256], uint16_t *d_out, int symmetric, void *stream)
{
    int rc = iiv::check_mode(mode);
    if (rc) return rc;
    if ((rc = iiv::check_dm(mode, dm))) return rc;
    if (!d_out) return iiv::set_error(IIV_ERR_INVALID, "d_out is NULL");
    return iiv::build_table(mode, dm, d_out, symmetric, (hipStream_t)stream);
}

int iiv_build_store_table(int mode, const int32_t dm[256], uint16_t *d_out, void *stream)
{
    int rc = iiv::check_mode(mode);
    if (rc) return rc;
    if ((rc = iiv::check_dm(mode, dm))) return rc;
    if (!d_out) return iiv::set_error(IIV_ERR_INVALID, "d_out is NULL");
    return iiv::build_store_table(mode, dm, d_out, (hipStream_t)stream);
}

int iiv_symmetrise_table(int mode, uint16_t *d_table, void *stream)
{
    int rc = iiv::check_mode(mode);
    if (rc) return rc;
    if (!d_table) return iiv::set_error(IIV_ERR_INVALID, "d_table is NULL");
    return iiv::symmetrise_table(mode, d_table, (hipStream_t)stream);
}

int iiv_store_table_from_table(int mode, const uint16_t *d_table, uint16_t *d_store_out, void *stream)
{
    int rc = iiv::check_mode(mode);
    if (rc) return rc;
    if (!d_table || !d_store_out) return iiv::set_error(IIV_ERR_INVALID, "NULL table");
    return iiv::store_table_from_table(mode, d_table, d_store_out, (hipStream_t)stream);
}

int iiv_pack(int mode, int n, const uint8_t *d_main, const uint8_t *d_aux, uint64_t *d_packed, void *stream)
{
    int rc = iiv::check_mode(mode);
    if (rc) return rc;
    if (n < 0 || !d_main || !d_packed || (mode == IIV_DHGR && !d_aux))
        return iiv::set_error(IIV_ERR_INVALID, "iiv_pack: bad argument");
    return iiv::pack(mode, n, d_main, d_aux, d_packed, (hipStream_t)stream);
}

int iiv_diff_weights(int mode, const uint16_t *d_table, int n, const uint64_t *d_src_packed,
                     const uint64_t *d_tgt_packed, int is_aux, int32_t *d_out, void *stream)
{
    int rc = iiv::check_mode(mode);
    if (rc) return rc;
    if (n < 0 || !d_table || !d_src_packed || !d_tgt_packed || !d_out || (mode == IIV_HGR && is_aux))
        return iiv::set_error(IIV_ERR_INVALID, "iiv_diff_weights: bad argument");
    return iiv::diff_weights(mode, d_table, n, d_src_packed, d_tgt_packed, is_aux ? 1 : 0, d_out,
                             (hipStream_t)stream);
}

int iiv_compute_delta_pages(int mode, const uint16_t *d_table, int n, const uint64_t *d_tgt_packed,
                            const int32_t *d_pages, const int32_t *d_contents, const int32_t *d_dw_rows,
                            int is_aux, int32_t *d_out, void *stream)
{
    int rc = iiv::check_mode(mode);
    if (rc) return rc;
    if (n < 0 || !d_table || !d_tgt_packed || !d_pages || !d_contents || !d_dw_rows || !d_out ||
        (mode == IIV_HGR && is_aux))
        return iiv::set_error(IIV_ERR_INVALID, "iiv_compute_delta_pages: bad argument");
    return iiv::compute_delta_pages(mode, d_table, n, d_tgt_packed, d_pages, d_contents, d_dw_rows,
                                    is_aux ? 1 : 0, d_out, (hipStream_t)stream);
}

}  // extern "C"

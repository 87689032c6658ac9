"""ctypes binding of libthunder_amd.so -- the C ABI declared in include/thunder_amd.h.

There is no CPU fallback: importing this module without the built HIP library raises, and every call
raises ThxError on a non-zero status (the reference's own style is REPORT_ERROR + abort).
torch is used only as a device-memory / stream provider (tensor.data_ptr()).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("THX_LIB") or os.path.join(_HERE, "lib", "libthunder_amd.so")   # THX_LIB: an alternative build, for A/B runs of compile-time parameters


class ThxError(RuntimeError):
    pass


class CtfAttr(C.Structure):
    """struct CTFAttr, include/Database.h:302-330 (reference) == thx_ctf_attr."""
    _fields_ = [("voltage", C.c_float), ("defocusU", C.c_float), ("defocusV", C.c_float),
                ("defocusTheta", C.c_float), ("Cs", C.c_float), ("amplitudeContrast", C.c_float),
                ("phaseShift", C.c_float)]


class PfCtx(C.Structure):
    """thx_pf_ctx (include/thunder_amd.h)"""
    _fields_ = [("symQuat", C.c_void_p), ("nSym", C.c_int), ("img0", C.c_uint)]


class RefineConfig(C.Structure):
    """thx_refine_config (include/thunder_amd.h)"""
    _fields_ = [("N", C.c_int), ("pf", C.c_int), ("nImg", C.c_int), ("halfOfRank", C.c_int), ("nHalfA", C.c_int),
                ("mLR", C.c_int), ("mLT", C.c_int), ("nPhase", C.c_int), ("mReco", C.c_int), ("maxPhase", C.c_int),
                ("batch", C.c_int),
                ("rL", C.c_int), ("nGroup", C.c_int), ("groupSig", C.c_int), ("pixelOrder", C.c_int), ("wgPerCU", C.c_int),
                ("pixelSize", C.c_float), ("maskRadiusPx", C.c_float), ("sigma2Init", C.c_float),
                ("transS", C.c_double), ("transQ", C.c_double), ("pfL", C.c_double), ("pfS", C.c_double),
                ("peakFactorR", C.c_double), ("seed", C.c_ulonglong),
                ("coreFSC", C.c_int), ("goldenAverage", C.c_int), ("solventFlatten", C.c_int), ("normCorrection", C.c_int),
                ("nK", C.c_int), ("searchType", C.c_int), ("nR", C.c_int), ("nT", C.c_int), ("rScan", C.c_int), ("scanBatch", C.c_int),
                ("pfSGlobal", C.c_double), ("peakFactorC", C.c_double), ("scanMinK", C.c_double), ("scanMinS", C.c_double),
                ("balanceClass", C.c_int), ("nSym", C.c_int), ("symMat", C.c_void_p), ("symQuat", C.c_void_p),
                ("mLD", C.c_int), ("ctfRefineS", C.c_double), ("pfSCTF", C.c_double)]


SEARCH_LOCAL, SEARCH_GLOBAL, SEARCH_CTF = 0, 1, 2


class RefineCapture(C.Structure):
    """thx_refine_capture (include/thunder_amd.h): device pointers as integers"""
    _fields_ = [(n, C.c_void_p) for n in ("uR", "uT", "r", "t", "k123", "s01", "mapsFsc", "rP", "tP", "wRP", "wTP", "Fraw", "Traw",
                                          "scanUC", "scanUR", "scanUT", "r0", "t0", "k0", "s0", "Fsym", "Tsym", "uD", "dP", "dR")] + \
               [("phases", C.c_int)]


class RefineStats(C.Structure):
    """thx_refine_stats (include/thunder_amd.h)"""
    _fields_ = [("expectMs", C.c_double), ("insertMs", C.c_double), ("expectLaunches", C.c_long), ("expectImages", C.c_long),
                ("insertLaunches", C.c_long), ("insertImages", C.c_long), ("stageMs", C.c_double * 8),
                ("balancingRounds", C.c_long), ("iterations", C.c_long), ("imagePhases", C.c_long), ("nPxl", C.c_int),
                ("nPxlM", C.c_int),
                ("batch", C.c_int), ("insertGroups", C.c_ulonglong), ("lastRounds", C.c_int * 4), ("normMedian", C.c_float),
                ("normRadius", C.c_float), ("scanMs", C.c_double), ("scanLaunches", C.c_long), ("scanImages", C.c_long),
                ("nPxlS", C.c_int), ("nK", C.c_int), ("classCount", C.c_int * 16), ("lastRoundsK", C.c_int * 64),
                ("balanced", C.c_int * 16)]


class RefineView(C.Structure):
    """thx_refine_view (include/thunder_amd.h): device pointers as integers"""
    _fields_ = [(n, C.c_int) for n in ("nImg", "nPxl", "nPxlM", "nVol", "vdim", "rSig")] + \
               [(n, C.c_void_p) for n in ("iCol", "iRow", "iPxl", "iSig", "iColM", "iRowM", "img", "datP", "ctfP", "sigRcpP",
                                          "datM", "ctfM", "r", "t", "wR", "wT", "offset", "vols", "cells", "F", "T", "sig",
                                          "recoRot", "recoTran", "nP", "norm", "cls", "topR", "topT", "k123", "s01", "d", "wD",
                                          "maps", "mapsMAP")] + \
               [(n, C.c_int) for n in ("nK", "nPxlS", "fdim")]


_vp = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/thunder_amd.h one to one
SIGNATURES = {
    "thx_last_error": (C.c_char_p, []),
    "thx_version": (_i, []),
    "thx_device_count": (_i, [C.POINTER(_i)]),
    "thx_set_device": (_i, [_i]),
    "thx_malloc_dev": (_i, [C.POINTER(_vp), _sz]),
    "thx_free_dev": (_i, [_vp]),
    "thx_memcpy_h2d": (_i, [_vp, _vp, _sz]),
    "thx_memcpy_d2h": (_i, [_vp, _vp, _sz]),
    "thx_memset_dev": (_i, [_vp, _i, _sz]),
    "thx_device_sync": (_i, []),
    "thx_knobs_reload": (_i, []),
    "thx_comm_unique_id": (_i, [_vp]),
    "thx_comm_init": (_i, [C.POINTER(_vp), _vp, _i, _i]),
    "thx_comm_destroy": (_i, [_vp]),
    "thx_comm_rank": (_i, [_vp]),
    "thx_comm_size": (_i, [_vp]),
    "thx_comm_allreduce_f32": (_i, [_vp, _vp, _sz, _vp]),
    "thx_comm_allreduce_f64": (_i, [_vp, _vp, _sz, _vp]),
    "thx_comm_allreduce_i32": (_i, [_vp, _vp, _sz, _vp]),
    "thx_comm_allreduce_max_f64": (_i, [_vp, _vp, _sz, _vp]),
    "thx_comm_allreduce_i64": (_i, [_vp, _vp, _sz, _vp]),
    "thx_comm_reduce_i64": (_i, [_vp, _vp, _sz, _i, _vp]),
    "thx_comm_transport": (C.c_char_p, [_vp]),
    "thx_comm_broadcast": (_i, [_vp, _vp, _sz, _i, _vp]),
    "thx_reco_allreduce_workspace": (_sz, [_i, _i, _i]),
    "thx_reco_allreduce": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "thx_reco_sphere_pack_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, C.POINTER(C.c_long), _vp]),
    "thx_compare_hemispheres_dev": (_i, [_vp, _vp, _i, _i, _vp, _vp, _f, _f, _i, _i, C.c_ulonglong, C.c_uint, C.POINTER(_i),
                                         _vp]),
    "thx_core_mask_dev": (_i, [_vp, _i, _f, _f, _vp]),
    "thx_soft_mask_volume_dev": (_i, [_vp, _i, _f, _f, _f, _vp]),
    "thx_random_phase_dev": (_i, [_vp, _vp, _i, _i, C.c_ulonglong, C.c_uint, _vp, _vp]),
    "thx_thu_write": (_i, [C.c_char_p, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "thx_thu_load_extra": (_i, [C.c_char_p, _i, _vp, _i, _vp, _vp, _vp]),
    "thx_pixel_list_host": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, C.POINTER(_i)]),
    "thx_view_order_host": (_i, [_vp, _i, _vp]),
    "thx_draw_reco_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_ulonglong, C.c_uint, C.c_uint, _vp]),
    "thx_refine_create": (_i, [C.POINTER(_vp), C.POINTER(RefineConfig), _vp, _vp]),
    "thx_refine_destroy": (_i, [_vp]),
    "thx_refine_set_particles": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "thx_refine_set_reference": (_i, [_vp, _vp, _vp]),
    "thx_refine_set_image_base": (_i, [_vp, C.c_longlong]),
    "thx_refine_reset": (_i, [_vp, _vp]),
    "thx_refine_iterate": (_i, [_vp, _vp, _i, _vp]),
    "thx_refine_get_map": (_i, [_vp, _i, _vp, _vp]),
    "thx_refine_set_capture": (_i, [_vp, _vp]),
    "thx_refine_get_state": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "thx_refine_get_stats": (_i, [_vp, C.POINTER(RefineStats), _i]),
    "thx_refine_get_view": (_i, [_vp, C.POINTER(RefineView)]),
    "thx_refine_get_map_k": (_i, [_vp, _i, _i, _vp, _vp]),
    "thx_refine_set_classes": (_i, [_vp, _vp, _vp]),
    "thx_refine_set_grid": (_i, [_vp, _vp, _vp, _vp]),
    "thx_refine_set_search_type": (_i, [_vp, _i]),
    "thx_refine_set_cutoff": (_i, [_vp, _i, _i, _vp]),
    "thx_refine_get_cutoff": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "thx_reco_allreduce_acc_class": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "thx_reco_reduce_acc_class": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "thx_rotmat_dev": (_i, [_vp, _vp, _i, _vp]),
    "thx_translate_dev": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "thx_ctf_dev": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _vp]),
    "thx_expect_precal_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _i, _i, _vp]),
    "thx_ctf_dsearch_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "thx_ctf_image_dev": (_i, [_vp, _vp, _f, _i, _i, _vp]),
    "thx_gather_pixels_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "thx_project_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "thx_logdatavsprior_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "thx_expect_local_workspace": (_sz, [_i, _i, _i, _i]),
    "thx_expect_local_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "thx_expect_local_packed_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "thx_projector_packed_bytes": (_sz, [_i]),
    "thx_projector_pack_dev": (_i, [_vp, _vp, _i, _i, _vp]),
    "thx_expect_global_workspace": (_sz, [_i, _i, _i]),
    "thx_expect_global_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i,
                                   _vp, _vp]),
    "thx_insert_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp,
                            _vp, _i, _i, _i, _i, _i, _vp]),
    "thx_insert_acc_bytes": (_sz, [_i, _i]),
    "thx_insert_bounds_dev": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "thx_insert_scale_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, C.c_long, _vp, _vp]),
    "thx_insert_accumulate_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f,
                                       _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "thx_insert_finish_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "thx_insert_groups_total": (_i, [_vp, _i, _vp]),
    "thx_release_stream": (_i, [_vp]),
    "thx_reco_allreduce_acc_workspace": (_sz, [_i, _i, _i]),
    "thx_reco_allreduce_acc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "thx_normalise_tf_dev": (_i, [_vp, _vp, _i, _vp]),
    "thx_symmetrize_dev": (_i, [_vp, _vp, _i, _i, _vp, _i, _d, _vp]),
    "thx_reco_create": (_i, [C.POINTER(_vp), _i, _i, _i, _f, _f]),
    "thx_reco_destroy": (_i, [_vp]),
    "thx_reco_set_balance_rounds": (_i, [_vp, _i, _i]),
    "thx_reco_floor_T_dev": (_i, [_vp, _vp, _i, _vp]),
    "thx_reco_reconstruct_dev": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, C.POINTER(_i), C.POINTER(_f),
                                      _vp]),
    "thx_reco_set_projectee_dev": (_i, [_vp, _vp, _vp, _vp]),
    "thx_fft3d_fw_dev": (_i, [_vp, _vp, _i, _vp]),
    "thx_fft3d_bw_dev": (_i, [_vp, _vp, _i, _vp]),
    "thx_fsc_dev": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "thx_remask_dev": (_i, [_vp, _i, _i, _f, _f, _vp]),
    "thx_translate_image_dev": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "thx_translate_volume_dev": (_i, [_vp, _vp, _i, _f, _d, _d, _d, _vp]),
    "thx_sigma_spectra_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i,
                                   _vp]),
    "thx_sigma_spectra_packed_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i,
                                   _vp]),
    "thx_sigma_accum_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "thx_sigma_final_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _vp]),
    "thx_mrc_info": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "thx_reco_reconstruct_async_dev": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "thx_mrc_read_images": (_i, [C.c_char_p, _i, _i, _vp]),
    "thx_mrc_read_volume": (_i, [C.c_char_p, _vp]),
    "thx_mrc_write_volume": (_i, [C.c_char_p, _vp, _i, _i, _i, _f]),
    "thx_mrc_write_stack": (_i, [C.c_char_p, _vp, _i, _i, _f]),
    "thx_thu_count": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i)]),
    "thx_thu_load": (_i, [C.c_char_p, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "thx_img_subtract_bg_dev": (_i, [_vp, _i, _i, _f, _vp]),
    "thx_img_stats_dev": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "thx_img_mask_normalise_fft_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _vp]),
    "thx_pf_perturb_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _d, _d, _d, _d, C.c_ulonglong, C.c_uint, _vp, _vp]),
    "thx_pf_update_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _d, C.c_ulonglong,
                               C.c_uint, _vp, _vp]),
    "thx_symmetry_host": (_i, [C.c_char_p, _vp, _vp, _i, C.POINTER(_i)]),
    "thx_pf_symmetrise_dev": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp]),
    "thx_pf_cal_vari_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, C.c_ulonglong, C.c_uint, _vp, _vp]),
    "thx_pf_perturb_ex_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _d, _d, _d, _d, C.c_ulonglong, C.c_uint, _vp, _vp, _vp]),
    "thx_pf_update_ex_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _d, C.c_ulonglong,
                                  C.c_uint, _vp, _vp, _vp]),
    "thx_pf_perturb_d_ex_dev": (_i, [_vp, _vp, _vp, _i, _i, _d, _i, C.c_ulonglong, C.c_uint, _vp, C.c_uint, _vp]),
    "thx_pf_update_d_ex_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, C.c_ulonglong, C.c_uint, _vp, C.c_uint, _vp]),
    "thx_pf_class_select_ex_dev": (_i, [_vp, _vp, _vp, _i, _i, _d, C.c_ulonglong, C.c_uint, C.c_uint, _vp]),
    "thx_pf_scan_support_ex_dev": (_i, [_vp] * 13 + [_i] * 6 + [_d, _d, _d, C.c_ulonglong, C.c_uint, _vp, _vp]),
    "thx_pf_stop_init_dev": (_i, [_vp, _vp, _vp, _d, _d, _i, _vp]),
    "thx_pf_stop_rule_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "thx_pf_acg_stats_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "thx_pf_perturb_d_dev": (_i, [_vp, _vp, _vp, _i, _i, _d, _i, C.c_ulonglong, C.c_uint, _vp, _vp]),
    "thx_pf_update_d_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, C.c_ulonglong, C.c_uint, _vp, _vp]),
    "thx_pf_class_select_dev": (_i, [_vp, _vp, _vp, _i, _i, _d, C.c_ulonglong, C.c_uint, _vp]),
    "thx_norm_residual_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp]),
    "thx_norm_residual_packed_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp]),
    "thx_median_f32_dev": (_i, [_vp, _vp, _i, _vp]),
    "thx_norm_scale_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "thx_pf_scan_support_dev": (_i, [_vp] * 13 + [_i] * 5 + [_d, _d, _d, C.c_ulonglong, C.c_uint, _vp]),
    "thx_ExpectProject_host": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "thx_ExpectRotran_host": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "thx_InsertFT_host": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i,
                               _i, _i, _i, _i, _i, _i, _i]),
    "thx_InsertFT_hemi_host": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i,
                               _i, _i, _i, _i, _i, _i, _i]),
    "thx_PrepareTF_host": (_i, [_i, _vp, _vp, _i, _vp, _i, _i, _i]),
    "thx_ReconstructG_host": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _i, _i, _i, _i, _vp]),
    "thx_ExpectPrecal_host": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i]),
    "thx_ExpectGlobal3D_host": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "thx_GCTFinit_host": (_i, [_vp, _vp, _f, _i, _i]),
    # per-image / per-stage Interface.h entries (thx_iface.hip, staged reconstructG in thx_reco.hip)
    "thx_texture_create": (_i, [C.POINTER(_vp), _i, _i, _i]),
    "thx_texture_destroy": (_i, [_vp]),
    "thx_texture_device": (_i, [_vp]),
    "thx_calpoint_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i]),
    "thx_calpoint_destroy": (_i, [_vp]),
    "thx_ExpectPreidx_host": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _i]),
    "thx_ExpectPrefre_host": (_i, [_i, C.POINTER(_vp), _vp, _i]),
    "thx_ExpectLocalIn_host": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i]),
    "thx_ExpectLocalV3D_host": (_i, [_i, _vp, _vp, _i]),
    "thx_ExpectLocalP_host": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "thx_ExpectLocalHostA_host": (_i, [_i] + [C.POINTER(_vp)] * 10 + [_i, _i, _i, _i]),
    "thx_ExpectLocalRTD_host": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "thx_ExpectLocalPreI3D_host": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i, _i, _i, _i, _i]),
    "thx_ExpectLocalM_host": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i]),
    "thx_ExpectLocalHostF_host": (_i, [_i] + [C.POINTER(_vp)] * 10 + [_i]),
    "thx_ExpectLocalFin_host": (_i, [_i] + [C.POINTER(_vp)] * 5 + [_i]),
    "thx_ExpectFreeIdx_host": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp)]),
    "thx_ExposePT_host": (_i, [_i, _vp, _i, _i, _i, _vp, _i, _i, _i]),
    "thx_ExposeWT_host": (_i, [_i, _vp, _vp, _vp, _i, _f, _i, _i, _i, _i, _i, _i]),
    "thx_ExposeWT_plain_host": (_i, [_i, _vp, _vp, _i, _i, _i]),
    "thx_AllocDevicePoint_host": (_i, [_i] + [C.POINTER(_vp)] * 7 + [C.POINTER(_vp), _i, _i, _i]),
    "thx_HostDeviceInit_host": (_i, [_i, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "thx_ExposeC_host": (_i, [_i, _vp, _vp, _vp, _vp, C.POINTER(_vp), _i, _i]),
    "thx_ExposeForConvC_host": (_i, [_i, _vp, _vp, _vp, C.POINTER(_vp), _f, _f, _i, _i, _i, _i, _i]),
    "thx_ExposeWC_host": (_i, [_i, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_f), _i, _i, _i, _i]),
    "thx_FreeDevHostPoint_host": (_i, [_i] + [C.POINTER(_vp)] * 7 + [C.POINTER(_vp), _vp, _i, _i]),
    "thx_ExposePFW_host": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i]),
    "thx_ExposePF_host": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "thx_ExposeCorrF_host": (_i, [_i, _vp, _vp, _f, _i]),
    "thx_ExposeCorrF_fft_host": (_i, [_i, _vp, _vp, _vp, _f, _i]),
    "thx_ReMask_host": (_i, [_vp, _f, _f, _f, _i, _i]),
    "thx_TranslateI2D_host": (_i, [_i, _vp, _d, _d, _i, _i]),
    "thx_TranslateI_host": (_i, [_i, _vp, _d, _d, _d, _i, _i]),
}

_lib = None


def load(path=None):
    """dlopen the HIP library (import torch first so its bundled libamdhip64 / hipFFT are the ones bound)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ThxError("libthunder_amd.so is not built (%s missing): run `python -m thunder_amd.build`; "
                       "there is no CPU fallback for the hot path" % path)
    try:
        import torch  # noqa: F401  (loads the ROCm runtime torch ships, so both share one HIP context)
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI symbol missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise ThxError("thunder_amd: %s (status %d)" % (load().thx_last_error().decode(errors="replace"), rc))


def ptr(x):
    """device/host pointer of a torch tensor, numpy array, int or None"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous(), "tensor must be contiguous"
        return x.data_ptr()
    if hasattr(x, "ctypes"):
        assert x.flags.c_contiguous, "array must be contiguous"
        return x.ctypes.data
    raise TypeError(type(x))


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream


def call(name, *args):
    check(getattr(load(), name)(*args))

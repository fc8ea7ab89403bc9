"""Thin Python host layer over the C ABI: context, device buffers and the batched calls.

Plumbing only -- every computation happens in libvtmhip.so.  numpy arrays go in/out through
vtmhip_dev_alloc / vtmhip_h2d / vtmhip_d2h; torch users pass `tensor.data_ptr()` wherever a device pointer is
expected and `torch.cuda.current_stream().cuda_stream` to `Context.set_stream`."""
import ctypes as C

import numpy as np

from . import lib as _lib
from .lib import (AffineJob, DistJob, FracJob, FracResult, FullJob, IfJob, McJob, MeResult, PelOpJob, PicParams, QuantJob,   # noqa: F401
                  TrJob, TuJob, TuResult, TzJob, VtmHipError)


class DevBuf:
    """A device allocation owned by a Context (freed with the context or explicitly)."""

    def __init__(self, ctx, nbytes, dtype=np.uint8, shape=None):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.dtype = np.dtype(dtype)
        self.shape = shape
        p = C.c_void_p()
        ctx._check(ctx.L.vtmhip_dev_alloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value
        ctx._bufs.append(self)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._keep.append(arr)   # the copy is asynchronous: keep the source alive until the next sync
        self.ctx._check(self.ctx.L.vtmhip_h2d(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def to_host(self, dtype=None, shape=None):
        dtype = np.dtype(dtype or self.dtype)
        out = np.empty(self.nbytes // dtype.itemsize, dtype=dtype)
        self.ctx._check(self.ctx.L.vtmhip_d2h(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        self.ctx._keep.clear()
        shape = shape or self.shape
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr:
            self.ctx.L.vtmhip_dev_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """One per process / GPU (reference analogue: one RdCost/InterSearch stack, EncLib.cpp:110-122)."""

    def __init__(self, device=0):
        self.L = _lib.load()
        h = C.c_void_p()
        st = self.L.vtmhip_create(device, C.byref(h))
        if st != _lib.OK:
            raise VtmHipError(st, self.L.vtmhip_status_string(st).decode())
        self.h = h
        self._bufs = []
        self._keep = []

    def _check(self, st):
        if st != _lib.OK:
            raise VtmHipError(st, self.L.vtmhip_last_error(self.h).decode())

    def close(self):
        if self.h:
            for b in self._bufs:
                b.free()
            self.L.vtmhip_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_ptr):
        self._check(self.L.vtmhip_set_stream(self.h, stream_ptr))

    def use_own_stream(self):
        self._check(self.L.vtmhip_use_own_stream(self.h))

    def sync(self):
        self._check(self.L.vtmhip_sync(self.h))
        self._keep.clear()

    def alloc(self, nbytes, dtype=np.uint8, shape=None):
        return DevBuf(self, nbytes, dtype, shape)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DevBuf(self, max(arr.nbytes, 1), arr.dtype, arr.shape).upload(arr)

    def timer_start(self):
        self._check(self.L.vtmhip_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float()
        self._check(self.L.vtmhip_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    # ---- pointer-surface calls (host arrays) ------------------------------------------------------------------
    def xGetSAD(self, org, org_stride, cur, cur_stride, w, h, sub_shift=0, org_off=0, cur_off=0):
        d = C.c_uint64()
        self._check(self.L.vtmhip_xGetSAD(self.h, org.ctypes.data + 2 * org_off, org_stride, cur.ctypes.data + 2 * cur_off,
                                          cur_stride, w, h, sub_shift, C.byref(d)))
        return d.value

    def xGetHADs(self, org, org_stride, cur, cur_stride, w, h, org_off=0, cur_off=0):
        d = C.c_uint64()
        self._check(self.L.vtmhip_xGetHADs(self.h, org.ctypes.data + 2 * org_off, org_stride, cur.ctypes.data + 2 * cur_off,
                                           cur_stride, w, h, C.byref(d)))
        return d.value

    def xGetSSE(self, org, org_stride, cur, cur_stride, w, h, org_off=0, cur_off=0):
        d = C.c_uint64()
        self._check(self.L.vtmhip_xGetSSE(self.h, org.ctypes.data + 2 * org_off, org_stride, cur.ctypes.data + 2 * cur_off,
                                          cur_stride, w, h, C.byref(d)))
        return d.value

    def xGetSADwMask(self, org, org_stride, cur, cur_stride, w, h, mask, mask_off, mask_stride, step_x, mask_stride2, sub_shift=0):
        """DF_SAD_WITH_MASK on host arrays; `mask` is a 1-D int16 array, mask_off the index of DistParam::mask inside it."""
        d = C.c_uint64()
        self._check(self.L.vtmhip_xGetSADwMask(self.h, org.ctypes.data, org_stride, cur.ctypes.data, cur_stride, w, h, sub_shift,
                                               mask.ctypes.data + 2 * mask_off, mask_stride, step_x, mask_stride2, C.byref(d)))
        return d.value

    def masked_sad_batch(self, d_org, d_cur, d_mask, d_jobs, n, d_dist):
        self._check(self.L.vtmhip_masked_sad_batch_dev(self.h, d_org, d_cur, d_mask, d_jobs, n, d_dist))

    def weightedGeoBlk(self, src0, src1, w, h, weight, weight_off, step_x, weight_stride, bit_depth=10, clip=None):
        """m_weightedGeoBlk on host arrays (h x w int16 blocks); `weight` is a 1-D int16 array, weight_off the index of the first weight."""
        clip = clip or (0, (1 << bit_depth) - 1)
        dst = np.zeros((h, w), np.int16)
        self._check(self.L.vtmhip_weightedGeoBlk(self.h, src0.ctypes.data, src0.strides[0] // 2, src1.ctypes.data, src1.strides[0] // 2, dst.ctypes.data, w,
                                                 w, h, weight.ctypes.data + 2 * weight_off, step_x, weight_stride, bit_depth, clip[0], clip[1]))
        return dst

    def weightedGeoBlk_batch(self, d_src, d_dst, d_weight, d_jobs, n, bit_depth=10, clip=None):
        clip = clip or (0, (1 << bit_depth) - 1)
        self._check(self.L.vtmhip_weightedGeoBlk_batch_dev(self.h, d_src, d_dst, d_weight, d_jobs, n, bit_depth, clip[0], clip[1]))

    def filter(self, vertical, taps, is_first, is_last, src, src_off, src_stride, w, h, coeff, bit_depth=10, clip=None, bimc=0):
        """m_filterHor/m_filterVer[taps][isFirst][isLast] on a host array; returns the h x w int16 block."""
        clip = clip or (0, (1 << bit_depth) - 1)
        dst = np.zeros((h, w), np.int16)
        co = np.ascontiguousarray(coeff, dtype=np.int16)
        fn = self.L.vtmhip_filterVer if vertical else self.L.vtmhip_filterHor
        self._check(fn(self.h, taps, is_first, is_last, src.ctypes.data + 2 * src_off, src_stride, dst.ctypes.data, w, w, h,
                       co.ctypes.data, bit_depth, clip[0], clip[1], bimc))
        return dst

    def filter_copy(self, is_first, is_last, src, src_off, src_stride, w, h, bit_depth=10, clip=None, bimc=0):
        clip = clip or (0, (1 << bit_depth) - 1)
        dst = np.zeros((h, w), np.int16)
        self._check(self.L.vtmhip_filterCopy(self.h, is_first, is_last, src.ctypes.data + 2 * src_off, src_stride, dst.ctypes.data,
                                             w, w, h, bit_depth, clip[0], clip[1], bimc))
        return dst

    def fastFwdTrans(self, ttype, n, src, shift, line, skip1, skip2):
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.zeros(n * line, np.int32)
        self._check(self.L.vtmhip_fastFwdTrans(self.h, ttype, n, src.ctypes.data, dst.ctypes.data, shift, line, skip1, skip2))
        return dst

    def fastInvTrans(self, ttype, n, src, shift, line, skip1, skip2, cmin=-32768, cmax=32767):
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.zeros(n * line, np.int32)
        self._check(self.L.vtmhip_fastInvTrans(self.h, ttype, n, src.ctypes.data, dst.ctypes.data, shift, line, skip1, skip2, cmin, cmax))
        return dst

    # ---- batched device calls (device pointers: DevBuf.ptr or tensor.data_ptr()) ---------------------------------
    def dist_batch(self, d_org, d_cur, d_jobs, n, d_out):
        self._check(self.L.vtmhip_dist_batch_dev(self.h, d_org, d_cur, d_jobs, n, d_out))

    def dist_uniform_batch(self, d_org, d_cur, d_jobs, n, kind, w, h, sub_shift, d_out):
        """Every job is w x h of one kind: small blocks share a wave (vtmhip_dist_uniform_batch_dev)."""
        self._check(self.L.vtmhip_dist_uniform_batch_dev(self.h, d_org, d_cur, d_jobs, n, kind, w, h, sub_shift, d_out))

    def satd8_grid(self, d_org, org_stride, d_ref, ref_stride, w, h, r, d_out):
        self._check(self.L.vtmhip_satd8_grid_dev(self.h, d_org, org_stride, d_ref, ref_stride, w, h, r, d_out))

    def if_batch(self, d_src, d_dst, d_jobs, n):
        self._check(self.L.vtmhip_if_batch_dev(self.h, d_src, d_dst, d_jobs, n))

    def frac_search_batch(self, d_org, d_ref, d_jobs, n, max_w, max_h, d_results, uniform_square=False):
        self._check(self.L.vtmhip_frac_search_batch_dev(self.h, d_org, d_ref, d_jobs, n, max_w, max_h, int(uniform_square), d_results))

    def xT_batch(self, d_resi, d_coef, d_jobs, n, max_w, max_h, d_sum_abs=None):
        self._check(self.L.vtmhip_xT_batch_dev(self.h, d_resi, d_coef, d_jobs, n, max_w, max_h, d_sum_abs))

    def xT_uniform_batch(self, d_resi, d_jobs, n, w, h, d_coef, d_results):
        """Forward transforms of n uniform w x h TUs (TuJob table): coefficients + sum|coef| (vtmhip_xT_uniform_batch_dev)."""
        self._check(self.L.vtmhip_xT_uniform_batch_dev(self.h, d_resi, d_jobs, n, w, h, d_coef, d_results))

    def xIT_batch(self, d_coef, d_resi, d_jobs, n, max_w, max_h):
        self._check(self.L.vtmhip_xIT_batch_dev(self.h, d_coef, d_resi, d_jobs, n, max_w, max_h))

    def quant_batch(self, d_coef, d_q, d_delta_u, d_jobs, n, d_abs_sum):
        self._check(self.L.vtmhip_quant_batch_dev(self.h, d_coef, d_q, d_delta_u, d_jobs, n, d_abs_sum))

    def dequant_batch(self, d_q, d_coef, d_jobs, n):
        self._check(self.L.vtmhip_dequant_batch_dev(self.h, d_q, d_coef, d_jobs, n))

    def full_search_batch(self, pic, d_org, d_ref, d_jobs, n, d_results, square=0, uniform=None):
        """square = S / uniform = (W, H): the caller promises S x S (W x H) jobs with searchRange <= 4 (lane-per-candidate kernel)."""
        if uniform:
            self._check(self.L.vtmhip_full_search_uniform_batch_dev(self.h, C.byref(pic), d_org, d_ref, d_jobs, n, uniform[0], uniform[1], d_results))
        elif square:
            self._check(self.L.vtmhip_full_search_square_batch_dev(self.h, C.byref(pic), d_org, d_ref, d_jobs, n, square, d_results))
        else:
            self._check(self.L.vtmhip_full_search_batch_dev(self.h, C.byref(pic), d_org, d_ref, d_jobs, n, d_results))

    def motion_estimation_batch(self, pic, cfg, d_org, d_ref, d_other, d_jobs, n, max_w, max_h, d_results):
        """InterSearch::xMotionEstimation for n (PU, list, refIdx) jobs (vtmhip_xMotionEstimation_batch_dev)."""
        self._check(self.L.vtmhip_xMotionEstimation_batch_dev(self.h, C.byref(pic), C.byref(cfg), d_org, d_ref, d_other, d_jobs, n,
                                                              max_w, max_h, d_results))

    def estimate_mvp_amvp_batch(self, pic, d_org, d_ref, d_jobs, n, max_w, max_h, uniform=False, add_idx_bits=True, d_dist_bip=None):
        """InterSearch::xEstimateMvPredAMVP (template cost of the AMVP candidates) for n MeJob rows, in place"""
        self._check(self.L.vtmhip_xEstimateMvPredAMVP_batch_dev(self.h, C.byref(pic), d_org, d_ref, d_jobs, n, max_w, max_h, int(uniform), int(add_idx_bits), d_dist_bip))

    def pis_run_picture(self, levels, n_levels, buffers, main_stream, side_streams):
        """the whole level-order chain of a picture in one native call (levels: ctypes array of PisLevelRun; side_streams: list of raw stream handles)"""
        arr = (C.c_void_p * max(1, len(side_streams)))(*side_streams)
        self._check(self.L.vtmhip_pis_run_picture(self.h, levels, n_levels, C.byref(buffers), C.c_void_p(main_stream), arr, len(side_streams)))

    def pred_inter_search_batch(self, level_run, buffers):
        """InterSearch::predInterSearch (translational part + SMVD block) of the n real PUs of `level_run` (a PisLevelRun with pis.candsGiven = 1) in one call"""
        self._check(self.L.vtmhip_predInterSearch_batch_dev(self.h, C.byref(level_run), C.byref(buffers)))

    def is_uniform_shape(self, w, h):
        return bool(self.L.vtmhip_is_uniform_shape(w, h))

    def affine_motion_estimation_batch(self, pic, d_org, d_ref, d_other, d_jobs, n, max_w, max_h, d_results, bcw=False):
        """InterSearch::xAffineMotionEstimation per AffineMeJob (one workgroup per job); bcw: the batch may hold bi jobs under a CU-level BCW weight of -2 (32-bit variant beside)"""
        f = self.L.vtmhip_xAffineMotionEstimation_bcw_batch_dev if bcw else self.L.vtmhip_xAffineMotionEstimation_batch_dev
        self._check(f(self.h, C.byref(pic), d_org, d_ref, d_other, d_jobs, n, max_w, max_h, d_results))

    def mts_select_batch(self, d_results, num_tu, cands, w, h, bit_depth, max_cand, d_test):
        """TrQuant::transformNxN( trModes ) pre-selection of every TU of a level from the sum |coef| of its candidates (runs of num_tu results per candidate)"""
        arr = (C.c_uint8 * 8)(*cands)
        self._check(self.L.vtmhip_mts_select_batch_dev(self.h, d_results, num_tu, len(cands), arr, w, h, bit_depth, 15, max_cand, d_test))

    def smvd_batch(self, pic, d_org, d_ref, d_jobs, n, max_w, max_h, op, uniform=False):
        """the SMVD block of predInterSearch per SmvdJob, in place: op 0 xGetSymmetricCost, 1 xSymmetricMotionEstimation, 2 symmvdCheckBestMvp, 3 the whole block;
        uniform: every job is exactly max_w x max_h (VTMHIP_SMVD_UNIFORM: the lane-per-tile kernel for 8x8 .. 16x16)"""
        fn = (self.L.vtmhip_xGetSymmetricCost_batch_dev, self.L.vtmhip_xSymmetricMotionEstimation_batch_dev, self.L.vtmhip_symmvdCheckBestMvp_batch_dev)
        if op < 3 and not uniform:
            self._check(fn[op](self.h, C.byref(pic), d_org, d_ref, d_jobs, n, max_w, max_h))
        else:
            self._check(self.L.vtmhip_smvd_batch_dev(self.h, C.byref(pic), d_org, d_ref, d_jobs, n, max_w, max_h, op | (0x100 if uniform else 0)))

    def pred_affine_blk_batch(self, pic, d_ref, d_dst, d_jobs, n, max_w, max_h):
        self._check(self.L.vtmhip_xPredAffineBlk_batch_dev(self.h, C.byref(pic), d_ref, d_dst, d_jobs, n, max_w, max_h))

    def lfnst_tu_batch(self, d_coef, d_jobs, n):
        """TrQuant::xFwdLfnst / xInvLfnst (gather, core multiply, scatter) in place on n TU coefficient blocks"""
        self._check(self.L.vtmhip_lfnst_tu_batch_dev(self.h, d_coef, d_jobs, n))

    def kernel_timing(self, enable):
        """HIP events around every launch of the main kernels, on the launch stream (vtmhip_kernel_timing)"""
        self._check(self.L.vtmhip_kernel_timing(self.h, int(enable)))

    def kernel_timing_read(self, kernel):
        """(total milliseconds, launches) of one kernel since kernel_timing(True)"""
        ms, n = C.c_double(), C.c_int()
        self._check(self.L.vtmhip_kernel_timing_read(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def merge_cand_satd_batch(self, pic, d_org, d_ref, d_pred, d_plain, n_plain, d_bdof, n_bdof, d_dmvr, n_dmvr, d_mvd, max_w, max_h, d_dist, uniform=False, use_satd=True):
        """merge-candidate SATD pre-selection (hook B10): predictions of the three candidate classes + Hadamard distortion"""
        self._check(self.L.vtmhip_merge_cand_satd_batch_dev(self.h, C.byref(pic), d_org, d_ref, d_pred, d_plain, n_plain, d_bdof, n_bdof, d_dmvr, n_dmvr, d_mvd, max_w, max_h,
                                                            int(uniform), int(use_satd), d_dist))

    def pis_stage(self, level, stage):
        self._check(self.L.vtmhip_pis_stage(self.h, C.byref(level), stage))

    def motion_compensation_batch(self, d_org, d_ref, d_pred, d_out, d_jobs, n, max_w, max_h):
        """InterPrediction::motionCompensation per PU (uni / bi + addAvg) with the fused residual / removeHighFreq epilogue."""
        self._check(self.L.vtmhip_motion_compensation_batch_dev(self.h, d_org, d_ref, d_pred, d_out, d_jobs, n, max_w, max_h))

    def bdof_batch(self, d_org, d_ref, d_pred, d_out, d_jobs, n, max_w, max_h):
        """xPredInterBi with bioApplied (BDOF) for bi-predicted luma PUs; same job table and epilogues as motion_compensation_batch."""
        self._check(self.L.vtmhip_bdof_batch_dev(self.h, d_org, d_ref, d_pred, d_out, d_jobs, n, max_w, max_h))

    def dmvr_batch(self, pic, d_org, d_ref, d_pred, d_out, d_jobs, n, max_w, max_h, d_mvd=0):
        """xProcessDMVR (luma) for bi-predicted merge PUs: refined prediction (+ epilogue) and the sub-PU vector differences."""
        self._check(self.L.vtmhip_dmvr_batch_dev(self.h, C.byref(pic), d_org, d_ref, d_pred, d_out, d_jobs, n, max_w, max_h, d_mvd))

    def dmvr_chroma_batch(self, pic, d_org, d_ref, d_pred, d_out, d_jobs, n, max_w, max_h, d_mvd):
        """One 4:2:0 chroma plane of the PUs of a dmvr_batch call (jobs address that plane; d_mvd from the luma call)."""
        self._check(self.L.vtmhip_dmvr_chroma_batch_dev(self.h, C.byref(pic), d_org, d_ref, d_pred, d_out, d_jobs, n, max_w, max_h, d_mvd))

    def lfnst_set_tables(self, lfnst8x8, lfnst4x4):
        """The caller's LFNST core matrices (int8 [4][2][16][48] and [4][2][16][16]), once per context."""
        a, b = np.ascontiguousarray(lfnst8x8, np.int8), np.ascontiguousarray(lfnst4x4, np.int8)
        assert a.size == 4 * 2 * 16 * 48 and b.size == 4 * 2 * 16 * 16
        self._check(self.L.vtmhip_lfnst_set_tables(self.h, a.ctypes.data, b.ctypes.data))

    def lfnst(self, inverse, src, mode, index, size, zero_out):
        """TrQuant::fwdLfnstNxN / invLfnstNxN on a host vector; returns the trSize outputs."""
        src = np.ascontiguousarray(src, np.int32)
        dst = np.zeros(48 if size > 4 else 16, np.int32)
        fn = self.L.vtmhip_invLfnstNxN if inverse else self.L.vtmhip_fwdLfnstNxN
        self._check(fn(self.h, src.ctypes.data, dst.ctypes.data, mode, index, size, zero_out))
        return dst

    def lfnst_batch(self, d_src, d_dst, d_jobs, n):
        self._check(self.L.vtmhip_lfnst_batch_dev(self.h, d_src, d_dst, d_jobs, n))

    def mc_batch(self, d_ref, d_dst, d_jobs, n, max_w, max_h):
        """xPredInterBlk for luma and 4:2:0 chroma blocks (McJob.chroma)."""
        self._check(self.L.vtmhip_mc_batch_dev(self.h, d_ref, d_dst, d_jobs, n, max_w, max_h))

    def mc_luma_batch(self, d_ref, d_dst, d_jobs, n, max_w, max_h):
        self._check(self.L.vtmhip_mc_luma_batch_dev(self.h, d_ref, d_dst, d_jobs, n, max_w, max_h))

    def remove_high_freq_batch(self, d_org, d_pred, d_dst, d_jobs, n):
        self._check(self.L.vtmhip_remove_high_freq_batch_dev(self.h, d_org, d_pred, d_dst, d_jobs, n))

    def subtract_batch(self, d_a, d_b, d_dst, d_jobs, n):
        self._check(self.L.vtmhip_subtract_batch_dev(self.h, d_a, d_b, d_dst, d_jobs, n))

    def add_avg_batch(self, d_a, d_b, d_dst, d_jobs, n):
        self._check(self.L.vtmhip_add_avg_batch_dev(self.h, d_a, d_b, d_dst, d_jobs, n))

    def remove_weight_high_freq_batch(self, d_org, d_pred, d_dst, d_jobs, n):
        """BCW bi-pred ME target (PelOpJob.bcwWeight = weight of the searched list)."""
        self._check(self.L.vtmhip_remove_weight_high_freq_batch_dev(self.h, d_org, d_pred, d_dst, d_jobs, n))

    def add_weighted_avg_batch(self, d_a, d_b, d_dst, d_jobs, n):
        """BCW bi-prediction average (PelOpJob.bcwWeight = the list-1 weight)."""
        self._check(self.L.vtmhip_add_weighted_avg_batch_dev(self.h, d_a, d_b, d_dst, d_jobs, n))

    def tu_chain_batch(self, d_resi, d_jobs, n, max_w, max_h, d_results, d_levels=None, d_rec=None, uniform=False):
        self._check(self.L.vtmhip_tu_chain_batch_dev(self.h, d_resi, d_jobs, n, max_w, max_h, int(uniform), d_levels, d_rec, d_results))

    def tu_ts_chain_batch(self, d_resi, d_jobs, n, w, h, d_results, d_levels=None, d_rec=None):
        """transform-skip candidates (TuJob.typeHor == 3) of one TU size"""
        self._check(self.L.vtmhip_tu_ts_chain_batch_dev(self.h, d_resi, d_jobs, n, w, h, d_levels, d_rec, d_results))

    def affine_sobel_batch(self, d_pred, d_deriv, d_jobs, n):
        self._check(self.L.vtmhip_affine_sobel_batch_dev(self.h, d_pred, d_deriv, d_jobs, n))

    def affine_equal_coeff_batch(self, d_resi, d_deriv, d_jobs, n, d_eq):
        self._check(self.L.vtmhip_affine_equal_coeff_batch_dev(self.h, d_resi, d_deriv, d_jobs, n, d_eq))

    def tz_search_batch(self, pic, d_org, d_ref, d_jobs, n, d_results):
        self._check(self.L.vtmhip_tz_search_batch_dev(self.h, C.byref(pic), d_org, d_ref, d_jobs, n, d_results))


def struct_array_to_numpy(arr):
    """ctypes array of Structures -> uint8 numpy view (for upload)."""
    return np.frombuffer(arr, dtype=np.uint8)

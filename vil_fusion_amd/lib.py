"""ctypes loader for libvilfusion_hip.so (the product). Fails loudly: there is no CPU fallback."""
import ctypes as C
import os
import subprocess
from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("VILF_SO") or os.path.join(CSRC, "libvilfusion_hip.so")      # VILF_SO: another build of the library (same-box A/B of a kernel change: tools/)
_lib = None

EXPORTED = [
    "vilf_default_options", "vilf_create", "vilf_destroy", "vilf_reset", "vilf_last_error", "vilf_version",
    "vilf_window_solve", "vilf_window_solve_group", "vilf_window_marginalize", "vilf_batch_upload", "vilf_batch_solve", "vilf_batch_rewind",
    "vilf_batch_marginalize", "vilf_batch_download", "vilf_batch_download_states", "vilf_batch_summaries", "vilf_synchronize", "vilf_wait_for", "vilf_set_async_upload", "vilf_set_profiling", "vilf_get_profile",
    "vilf_batch_newest_poses_device", "vilf_prior_export", "vilf_prior_import", "vilf_eval_projection", "vilf_eval_imu", "vilf_eval_imu_raw",
    "vilf_eval_lidar_between", "vilf_eval_projection_td", "vilf_eval_prior", "vilf_eval_edge", "vilf_eval_surf", "vilf_pose_plus", "vilf_se3_plus",
    "vilf_imu_preintegrate", "vilf_imu_preintegrate_batch", "vilf_visual_imu_alignment", "vilf_posegraph_optimize", "vilf_scan2map_init", "vilf_scan2map_step", "vilf_scan2map_get_map", "vilf_scan2map_set_pose",
    "vilf_scan2map_batch_create", "vilf_scan2map_batch_init", "vilf_scan2map_batch_set_scan", "vilf_scan2map_batch_step", "vilf_scan2map_batch_snapshot",
    "vilf_scan2map_batch_rewind", "vilf_scan2map_batch_copy_stream", "vilf_scan2map_batch_results", "vilf_scan2map_batch_get_map", "vilf_get_profile_scan2map", "vilf_get_profile_marginalize", "vilf_batch_marginalize_stats", "vilf_get_profile_large_window", "vilf_lidar_extract_features", "vilf_feature_depth",
    "vilf_comm_unique_id", "vilf_comm_create", "vilf_comm_destroy", "vilf_gather_poses", "vilf_gather_poses_handle", "vilf_comm_ranks", "vilf_get_stream", "vilf_comm_last_error",
]


class VilfError(RuntimeError):
    pass


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of the HIP library (cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", CSRC, "-j4"] + ([] if verbose else ["-s"]))
    res = os.path.join(CSRC, "kernel_resources.json")          # registers / spills / LDS per kernel from the code objects (bench.py quotes the window kernels)
    tool = os.path.join(os.path.dirname(_HERE), "tools", "kernel_resources.py")
    if os.path.exists(tool) and (not os.path.exists(res) or os.path.getmtime(res) < os.path.getmtime(SO_PATH)):
        # generated next to the .so and git-ignored like it (it travels to the GPU box with the snapshot); a failure of the tool is reported, never fatal
        r = subprocess.run(["python3", tool, "__none__"], check=False, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0 or verbose:
            print(f"[vil_fusion_amd.build] tools/kernel_resources.py rc={r.returncode}" + (": " + r.stdout.strip()[-400:] if r.returncode != 0 else ""))
    return SO_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise VilfError(f"{SO_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                        "There is no CPU fallback for the solve path.")
    L = C.CDLL(SO_PATH)
    vp = C.c_void_p
    dpp = C.POINTER(abi.c_double_p)
    L.vilf_version.restype = C.c_char_p
    L.vilf_last_error.restype = C.c_char_p
    L.vilf_last_error.argtypes = [vp]
    L.vilf_default_options.argtypes = [C.POINTER(abi.Options)]
    L.vilf_default_options.restype = None
    L.vilf_create.argtypes = [C.POINTER(abi.Options), C.c_int, vp, C.POINTER(vp)]
    L.vilf_destroy.argtypes = [vp]
    L.vilf_destroy.restype = None
    L.vilf_reset.argtypes = [vp]
    L.vilf_window_solve.argtypes = [vp, C.POINTER(abi.WindowIn), C.POINTER(abi.WindowOut)]
    L.vilf_window_solve_group.argtypes = [vp, C.c_int, C.POINTER(abi.WindowIn), C.POINTER(abi.WindowOut)]
    L.vilf_window_marginalize.argtypes = [vp]
    L.vilf_batch_upload.argtypes = [vp, C.c_int, C.POINTER(abi.WindowIn)]
    L.vilf_batch_solve.argtypes = [vp, C.c_int]
    L.vilf_batch_rewind.argtypes = [vp]
    L.vilf_batch_marginalize.argtypes = [vp, C.c_int]
    L.vilf_batch_download.argtypes = [vp, C.c_int, C.c_int, C.POINTER(abi.WindowOut)]
    L.vilf_batch_summaries.argtypes = [vp, C.c_int, C.c_int, C.POINTER(abi.Summary)]
    L.vilf_synchronize.argtypes = [vp]
    L.vilf_wait_for.argtypes = [vp, vp]
    L.vilf_set_async_upload.argtypes = [vp, C.c_int]
    L.vilf_set_profiling.argtypes = [vp, C.c_int]
    L.vilf_get_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.vilf_batch_newest_poses_device.argtypes = [vp, abi.c_double_p, vp]
    L.vilf_prior_export.argtypes = [vp, C.c_int, C.POINTER(abi.Prior)]
    L.vilf_prior_import.argtypes = [vp, C.c_int, C.POINTER(abi.Prior)]
    L.vilf_eval_projection.argtypes = [vp, dpp, abi.c_double_p, abi.c_double_p, abi.c_double_p, dpp]
    L.vilf_eval_imu.argtypes = [vp, dpp, C.POINTER(abi.ImuPreint), abi.c_double_p, dpp]
    L.vilf_eval_imu_raw.argtypes = [vp, dpp, C.POINTER(abi.ImuPreint), abi.c_double_p, dpp, abi.c_double_p]
    L.vilf_eval_lidar_between.argtypes = [vp, dpp, C.POINTER(abi.LidarConstraint), abi.c_double_p, dpp]
    L.vilf_eval_edge.argtypes = [vp, abi.c_double_p, abi.c_double_p, abi.c_double_p, abi.c_double_p, abi.c_double_p, abi.c_double_p]
    L.vilf_eval_surf.argtypes = [vp, abi.c_double_p, abi.c_double_p, abi.c_double_p, C.c_double, abi.c_double_p, abi.c_double_p]
    L.vilf_pose_plus.argtypes = [vp, abi.c_double_p, abi.c_double_p, abi.c_double_p]
    L.vilf_se3_plus.argtypes = [vp, abi.c_double_p, abi.c_double_p, abi.c_double_p]
    L.vilf_imu_preintegrate.argtypes = [C.POINTER(abi.ImuNoise), abi.c_double_p, abi.c_double_p, abi.c_double_p, abi.c_double_p, C.c_int,
                                        abi.c_double_p, abi.c_double_p, abi.c_double_p, C.POINTER(abi.ImuPreint)]
    L.vilf_scan2map_init.argtypes = [vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int]
    L.vilf_scan2map_step.argtypes = [vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(abi.Scan2MapResult)]
    L.vilf_scan2map_get_map.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    L.vilf_scan2map_set_pose.argtypes = [vp, abi.c_double_p, abi.c_double_p]
    fpp = C.POINTER(C.c_float)
    L.vilf_scan2map_batch_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.vilf_scan2map_batch_init.argtypes = [vp, C.c_int, fpp, C.c_int, fpp, C.c_int, abi.c_double_p, abi.c_double_p]
    L.vilf_scan2map_batch_set_scan.argtypes = [vp, C.c_int, fpp, C.c_int, fpp, C.c_int]
    L.vilf_scan2map_batch_step.argtypes = [vp, C.c_int]
    L.vilf_scan2map_batch_snapshot.argtypes = [vp]
    L.vilf_get_profile_scan2map.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.vilf_get_profile_marginalize.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.vilf_batch_marginalize_stats.argtypes = [vp, C.POINTER(C.c_int)]
    L.vilf_get_profile_large_window.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.vilf_scan2map_batch_rewind.argtypes = [vp]
    L.vilf_scan2map_batch_copy_stream.argtypes = [vp, C.c_int, C.c_int]
    L.vilf_scan2map_batch_results.argtypes = [vp, C.c_int, C.c_int, C.POINTER(abi.Scan2MapResult)]
    L.vilf_scan2map_batch_get_map.argtypes = [vp, C.c_int, C.c_int, fpp, C.c_int, C.POINTER(C.c_int)]
    _lib = L
    return L


def default_options():
    o = abi.Options()
    lib().vilf_default_options(C.byref(o))
    return o

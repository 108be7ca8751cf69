"""Loader for the in-tree HIP shared library ``libmaniac_hip.so`` (C ABI: include/maniac_gpu.h).

There is deliberately no fallback of any kind: if the library is missing, fails to load or
reports no HIP device, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmaniac_hip.so")
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["mgpu_engine.hip", "mgpu_launch.hip", "mgpu_lanes.hip", "mgpu_windows.hip", "mgpu_host_setup.cpp", "mgpu_comm.cpp"]
HEADERS = ["mgpu_kernels.h", "mgpu_kernels_common.h", "mgpu_kernels_pair.h", "mgpu_kernels_recip.h", "mgpu_kernels_windows.h",
           "mgpu_internal.h", "mgpu_engine.h"]

MGPU_OK = 0
ABI_VERSION = 2          # include/maniac_gpu.h MGPU_ABI_VERSION: the library must report exactly this
MGPU_MOVE, MGPU_CREATION, MGPU_DELETION, MGPU_NONE = 0, 1, 2, 3
KERNEL_PAIR, KERNEL_RECIP, KERNEL_COMMIT, KERNEL_SFACTOR = 0, 1, 2, 3

# every symbol include/maniac_gpu.h declares (tests check the library exports each of them)
EXPORTS = [
    "mgpu_last_error", "mgpu_abi_version", "mgpu_device_count", "mgpu_box_prepare", "mgpu_ewald_setup",
    "mgpu_ewald_kvectors", "mgpu_coulomb_table_eval", "mgpu_rng_seed_streams", "mgpu_rng_fill", "mgpu_host_prefetch", "mgpu_engine_create", "mgpu_engine_destroy", "mgpu_engine_get_ewald",
    "mgpu_engine_get_kvectors", "mgpu_replica_set_molecules", "mgpu_replica_get_molecules",
    "mgpu_replica_num_molecules", "mgpu_replica_copy", "mgpu_system_energy", "mgpu_init_structure_factor",
    "mgpu_get_structure_factor", "mgpu_set_structure_factor", "mgpu_structure_factor_add", "mgpu_pair_energy_candidates",
    "mgpu_recip_energy_candidates", "mgpu_self_energy", "mgpu_intra_energy_candidates",
    "mgpu_trial_energy_candidates", "mgpu_commit_candidates", "mgpu_trial_submit", "mgpu_trial_wait",
    "mgpu_commit_submit", "mgpu_lane_site_buffer", "mgpu_set_host_team", "mgpu_replica_set_frames", "mgpu_replica_get_frames", "mgpu_move_trial_submit", "mgpu_move_trial_decide_submit", "mgpu_gcmc_trial_decide_submit", "mgpu_trial_decide_wait", "mgpu_gcmc_trial_submit", "mgpu_gcmc_trial_wait", "mgpu_replica_replace_molecule",
    "mgpu_replica_set_num_molecules", "mgpu_chain_window_capacity", "mgpu_chain_window", "mgpu_chain_set_margin",
    "mgpu_chain_get_stats", "mgpu_chain_set_timing", "mgpu_chain_get_timing", "mgpu_farm_window_capacity", "mgpu_farm_window_submit",
    "mgpu_farm_window_wait", "mgpu_farm_window_flush", "mgpu_farm_window_get_stats", "mgpu_comm_unique_id", "mgpu_comm_create",
    "mgpu_comm_destroy", "mgpu_comm_rank", "mgpu_allgather_block_stats", "mgpu_append_atom_records", "mgpu_format_fixed", "mgpu_phase_factors", "mgpu_synchronize", "mgpu_profile_enable", "mgpu_profile_reset",
    "mgpu_profile_get",
]


class MgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"maniac_gpu error {code}: {msg}")
        self.code = code


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP engine for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))]
    deps.append(os.path.join(_HERE, "..", "include", "maniac_gpu.h"))
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
            return LIB_PATH
    # RCCL (librccl, /opt/rocm/lib) carries the path's one collective (mgpu_comm.cpp)
    # -fopenmp: the host-side candidate loops of submit / wait / commit can use the caller's OpenMP team (libomp, as amdflang's)
    # the translation units are compiled side by side (each instantiates the kernels it launches), then linked
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(_HERE, "..", "build", "hip_obj")
    os.makedirs(obj_dir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fopenmp"]

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        cmd = ["hipcc"] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), max(1, (os.cpu_count() or 2) // 2))) as pool:
        objs = list(pool.map(compile_one, srcs))
    cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-fopenmp", "-o", LIB_PATH] + objs + \
          ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


def source_digest() -> str:
    """sha256 over the library's sources (csrc/*, include/maniac_gpu.h) in a fixed order: what a profile of the library
    belongs to, whatever the linker's build id or the output path did to the binary's own hash."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h")))
    files.append(os.path.join(_HERE, "..", "include", "maniac_gpu.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


_lib = None


def lib():
    """The loaded library.  Raises if it is absent -- never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.mgpu_last_error.restype = C.c_char_p
        L.mgpu_abi_version.restype = C.c_int
        if L.mgpu_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports ABI version {L.mgpu_abi_version()}, this package binds version {ABI_VERSION}: "
                               "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        for name in EXPORTS:
            if name in ("mgpu_last_error", "mgpu_host_prefetch"):
                continue
            getattr(L, name).restype = C.c_int          # a missing export raises here, not at first use
        _lib = L
    return _lib


def check(rc: int):
    if rc != MGPU_OK:
        raise MgpuError(rc, lib().mgpu_last_error().decode())

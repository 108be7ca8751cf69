"""Build / load the Fortran host side (maniac_mc_amd/fortran/*.f90 -> libmaniac_host.so) and drive it.

The Metropolis loop of the replica farm is Fortran (mc_farm.f90), as the north star asks: it calls
the HIP engine through the ISO_C_BINDING module maniac_gpu.f90 -> include/maniac_gpu.h.  Python only
creates the engine, hands over the initial configuration and asks for n steps.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import _lib
from .engine import Engine
from .system import System

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmaniac_host.so")
FSRC = [os.path.join(_HERE, "fortran", f) for f in ("maniac_gpu.f90", "mc_farm.f90", "maniac_output.f90", "mc_chain.f90")]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force: bool = False, verbose: bool = False) -> str:
    _lib.build()
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(f) for f in FSRC + [_lib.LIB_PATH]):
            return LIB_PATH
    moddir = os.path.join(_HERE, "..", "build", "fmod")
    os.makedirs(moddir, exist_ok=True)
    cmd = ["amdflang", "-O2", "-fopenmp", "-fPIC", "-shared", "-module-dir", moddir, "-o", LIB_PATH] + FSRC + \
          ["-L" + _HERE, "-lmaniac_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib/llvm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_host = None


def lib():
    global _host
    if _host is None:
        # the farm drives each lane from its own host thread with a nested OpenMP team: allow two active levels and
        # keep the inner teams alive between regions (must be in the environment before the OpenMP runtime starts)
        os.environ.setdefault("OMP_MAX_ACTIVE_LEVELS", "2")
        os.environ.setdefault("KMP_HOT_TEAMS_MAX_LEVEL", "2")
        os.environ.setdefault("KMP_HOT_TEAMS_MODE", "1")
        _lib.lib()       # libmaniac_hip.so first, so the dependency resolves in-tree
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (needs amdflang)")
        L = C.CDLL(LIB_PATH)
        L.mfarm_create.restype = C.c_int
        L.mfarm_run.restype = C.c_int
        L.mfarm_set_gcmc.restype = C.c_int
        _host = L
    return _host


class FortranFarm:
    """R chains of `system` on one GPU, advanced by the Fortran driver (mc_farm.f90).

    NVT by default (translation / rotation).  ``gcmc=dict(p_translation=..., p_rotation=..., fugacity=...)``
    switches insertion / deletion on: ``fugacity`` in molecules per cubic Angstrom, scalar, per active
    type, or (n_active, R) for an isotherm sweep; ``mol_capacity`` bounds the molecule count per type.
    """

    _live = None          # the farm created last that is still open (tests close leftovers through it)
    _slots = {}           # farm slot of the Fortran module (mfarm_select) -> the open farm that holds it
    MAX_FARMS = 8

    def __init__(self, system: System, n_replicas: int, device: int = 0, seed: int = 1,
                 translation_step: float = 0.3, rotation_step: float = 0.3, p_translation: float = 0.5,
                 rng_kind: int = 1, n_threads: int = 8, mol_capacity=None, gcmc=None,
                 n_lanes: int = 2, n_drivers: int = 1, device_build: bool = False, device_accept: bool = False,
                 window: bool = False, window_depth: int = 2):
        self.H = lib()
        # mc_farm.f90 keeps up to MAX_FARMS farms; every call below selects this farm's slot first
        free = [k for k in range(FortranFarm.MAX_FARMS) if k not in FortranFarm._slots]
        if not free:
            raise RuntimeError(f"{FortranFarm.MAX_FARMS} FortranFarms are already open in this process: close() one first")
        self.slot = -1
        self.sys = system
        self.R = int(n_replicas)
        topo = system.topo
        if mol_capacity is None:
            mol_capacity = [max(1, int(n)) for n in system.n_mol]
        self.mol_capacity = [int(c) for c in mol_capacity]
        self.eng = Engine(topo, system.box_matrix, system.bounds_lo, system.real_space_cutoff,
                          system.ewald_tolerance, self.R, device, self.mol_capacity)
        self.eng.load_system(system, 0)
        # device_build: the engine keeps the molecules' frames (com, offsets) and builds the trial moves itself; the
        # Fortran driver then holds no mirror of the coordinates (orthorhombic boxes)
        self.device_build = bool(device_build) and not system.is_triclinic()
        if self.device_build:
            for tt in range(topo.n_res):
                if topo.is_active[tt]:
                    self.eng.set_frames(0, tt, system.com[tt], system.offsets[tt])
        self.eng.init_structure_factor(0, True)
        for r in range(1, self.R):
            self.eng.replica_copy(r, 0)
        e0 = self.eng.system_energy(0)
        active = [t for t in range(topo.n_res) if topo.is_active[t]]
        self.active = np.array(active, dtype=np.int32)
        n1 = np.array([topo.atoms_in_res[t] for t in active], dtype=np.int32)
        nmol = np.array([system.n_mol[t] for t in active], dtype=np.int32)
        cap = np.array([self.mol_capacity[t] for t in active], dtype=np.int32)
        max_n1 = int(n1.max())
        tot = max(1, int(nmol.sum()))
        com = np.zeros((tot, 3))
        off = np.zeros((tot, max_n1, 3))
        pos = 0
        for k, t in enumerate(active):
            com[pos:pos + nmol[k]] = system.com[t]
            off[pos:pos + nmol[k], :n1[k]] = system.offsets[t]
            pos += nmol[k]
        energy0 = np.array([e0["non_coulomb"], e0["coulomb"], e0["recip_coulomb"], e0["ewald_self"],
                            e0["intra_coulomb"]])
        lo = np.ascontiguousarray(system.bounds_lo)
        length = np.ascontiguousarray(np.diag(system.box_matrix))
        self.slot = free[0]
        FortranFarm._slots[self.slot] = self
        self._select()
        # device_accept (with device_build): the engine also applies the acceptance rule and commits accepted candidates
        self.device_accept = bool(device_accept) and self.device_build
        # window (with device_build): one launch per lane step (mgpu_farm_window_submit): the engine builds, evaluates, decides
        # with the driver's draws and commits; the driver checks every decision.  Falls back to the batched path where the
        # one-launch kernel does not apply (mgpu_farm_window_capacity).
        want_window = bool(window) and self.device_build and not self.device_accept
        self.H.mfarm_configure(C.c_int((3 if want_window else (2 if self.device_accept else 1)) if self.device_build else 0))
        rc = self.H.mfarm_create(self.eng.h, C.c_int(self.R), C.c_int(len(active)), self.active.ctypes.data_as(_ip),
                                 n1.ctypes.data_as(_ip), nmol.ctypes.data_as(_ip), cap.ctypes.data_as(_ip),
                                 C.c_int(max_n1), com.ctypes.data_as(_dp), off.ctypes.data_as(_dp),
                                 energy0.ctypes.data_as(_dp), lo.ctypes.data_as(_dp), length.ctypes.data_as(_dp),
                                 C.c_double(system.temperature), C.c_double(translation_step),
                                 C.c_double(rotation_step), C.c_double(p_translation), C.c_int(seed),
                                 C.c_int(rng_kind), C.c_int(n_threads), C.c_int(n_lanes))
        _lib.check(rc)
        if system.is_triclinic():
            from .engine import box_prepare
            _, volume, reciprocal, _ = box_prepare(system.box_matrix)
            mat = np.asfortranarray(system.box_matrix, dtype=np.float64)
            rcp = np.asfortranarray(reciprocal, dtype=np.float64)
            self.H.mfarm_set_triclinic(mat.ctypes.data_as(_dp), rcp.ctypes.data_as(_dp), C.c_double(volume))
        self.H.mfarm_set_window_depth(C.c_int(int(window_depth)))
        mode = np.zeros(3)
        self.H.mfarm_window_mode(mode.ctypes.data_as(_dp))
        self.window = bool(mode[0])
        self.H.mfarm_set_drivers(C.c_int(max(1, int(n_drivers))))     # host threads that share the lanes (mc_farm.f90)
        self.n_drivers = max(1, int(n_drivers))
        self.max_n1 = max_n1
        self.n_lanes = max(1, min(int(n_lanes) if n_lanes > 0 else 2, 4, self.R))
        self.n_active = len(active)
        self.stats = np.zeros(3)
        FortranFarm._live = self
        if gcmc is not None:
            fug = np.asarray(gcmc["fugacity"], dtype=np.float64)
            if fug.ndim == 0:
                fug = np.full((self.R, self.n_active), float(fug))
            elif fug.ndim == 1:
                fug = np.tile(fug[None, :], (self.R, 1)) if fug.shape[0] == self.n_active and self.n_active > 1 \
                    else (np.tile(fug[:, None], (1, self.n_active)) if fug.shape[0] == self.R
                          else np.tile(fug[None, :], (self.R, 1)))
            fug = np.ascontiguousarray(fug.reshape(self.R, self.n_active))     # == Fortran (n_active, R)
            rc = self.H.mfarm_set_gcmc(C.c_double(gcmc["p_translation"]), C.c_double(gcmc["p_rotation"]),
                                       fug.ctypes.data_as(_dp))
            if rc:
                raise ValueError(f"mfarm_set_gcmc: bad probabilities / fugacity (code {rc})")

    def _select(self):
        self.H.mfarm_select(C.c_int(self.slot))

    def run(self, n_steps: int) -> int:
        """Advance every chain by n_steps move selections; returns the moves accepted during this call."""
        self._select()
        before = self.stats[1]
        rc = self.H.mfarm_run(C.c_int(n_steps), self.stats.ctypes.data_as(_dp))
        _lib.check(rc)
        return int(self.stats[1] - before)

    def window_mode(self):
        """(window mode on, windows of a lane in flight, steps the device left to the driver so far)."""
        self._select()
        mode = np.zeros(3)
        self.H.mfarm_window_mode(mode.ctypes.data_as(_dp))
        return bool(mode[0]), int(mode[1]), int(mode[2])

    def recalibrate(self):
        self._select()
        s = np.zeros(2)
        self.H.mfarm_recalibrate(s.ctypes.data_as(_dp))
        return s

    def timers(self):
        """Host seconds in: generate, submit, wait-for-GPU, resolve, commit-submit."""
        t = np.zeros(7)
        self._select()
        self.H.mfarm_get_timers(t.ctypes.data_as(_dp))
        return dict(zip(("generate", "submit", "wait", "resolve", "commit", "gen_rng", "gen_gather"), t.tolist()))

    def counters(self):
        c = np.zeros(8)
        self._select()
        self.H.mfarm_get_counters(c.ctypes.data_as(_dp))
        names = ("trial_translations", "translations", "trial_rotations", "rotations", "trial_creations",
                 "creations", "trial_deletions", "deletions")
        return dict(zip(names, c.astype(np.int64).tolist()))

    def counts(self):
        """Current molecule counts, shape (R, n_active)."""
        c = np.zeros((self.R, self.n_active), dtype=np.int32)
        self._select()
        self.H.mfarm_get_counts(c.ctypes.data_as(_ip))
        return c

    def exchange_block(self, comm, n_bins=5001):
        """mfarm_exchange_block: every rank's {accepted, trials} and per-active-type histogram of its chains' molecule
        counts through the C-ABI communicator `comm` (exchange.CAbiComm).  Returns (sums[world, 2], hist[world, n_active,
        n_bins])."""
        self._select()
        sums = np.zeros((comm.world, 2))
        hist = np.zeros((comm.world, self.n_active, n_bins), dtype=np.int64)
        self.H.mfarm_exchange_block.restype = C.c_int
        rc = self.H.mfarm_exchange_block(comm.h, C.c_int(n_bins), C.c_int(comm.world), sums.ctypes.data_as(_dp),
                                         hist.ctypes.data_as(C.POINTER(C.c_longlong)))
        _lib.check(rc)
        return sums, hist

    def energy(self, replica: int):
        e = np.zeros(5)
        self._select()
        self.H.mfarm_get_energy(C.c_int(replica), e.ctypes.data_as(_dp))
        return e

    def molecule(self, replica: int, ia: int, slot: int):
        com = np.zeros(3)
        off = np.zeros((self.max_n1, 3))
        self._select()
        self.H.mfarm_get_molecule(C.c_int(replica), C.c_int(ia), C.c_int(slot), com.ctypes.data_as(_dp),
                                  off.ctypes.data_as(_dp))
        return com, off

    @property
    def trials(self):
        return int(self.stats[0])

    @property
    def accepted(self):
        return int(self.stats[1])

    @property
    def skipped(self):
        return int(self.stats[2])

    def close(self):
        if FortranFarm._slots.get(getattr(self, "slot", -1)) is self:
            self._select()
            self.H.mfarm_destroy()
            del FortranFarm._slots[self.slot]
        if FortranFarm._live is self:
            FortranFarm._live = next(iter(FortranFarm._slots.values()), None)
        self.eng.close()

"""Host-side description of a GCMC system in the reference's own terms.

The reference keeps, per residue type ``t`` and molecule slot ``m``, a centre of mass
``primary%mol_com(:, t, m)`` and per-site offsets ``primary%site_offset(:, t, m, a)``
(/root/reference/src/simulation_state.f90:115-116); charges and atom types are functions
of ``(t, a)`` only (``primary%atom_charges(t, a)``, ``primary%atom_types(t, a)``,
simulation_state.f90:107-111) and the Lennard-Jones epsilon / sigma are assigned purely by
atom type (parameters_parser.f90:89-98, :141-176).  ``Topology`` / ``System`` hold exactly
that, in numpy, so the same object can be handed to the HIP engine, to the C restatement and
to the compiled reference.

Units are the reference's internal ones: Angstrom, elementary charge, Kelvin (E / k_B).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np

# /root/reference/src/constants.f90:7-20 -- kept bit-identical (same decimal literals)
PI = 3.14159265358979323846
TWOPI = 2.0 * PI
KB_KCALMOL = 0.0019872041
KB_JK = 1.380658e-23
EPS0_INV_EVA = 14.40198
KB_EVK = 8.6173852e-5
ERROR_TOL = 1.0e-10
# /root/reference/src/parameters.f90:8
NB_MAX_MOLECULE = 5000
# /root/reference/src/parameters.f90:28-29
A3_TO_M3 = 1.0e-30
ATM_TO_PA = 1.01325e5

# stated parity tolerance (BASELINE.json north_star): 1e-10 kcal/mol, in Kelvin
TOL_KCALMOL = 1.0e-10
TOL_K = TOL_KCALMOL / KB_KCALMOL


@dataclass
class Topology:
    """Residue types, their site templates and the force field."""

    atoms_in_res: np.ndarray          # (n_res,) int32
    atom_types: np.ndarray            # (n_res, max_atom) int32, 1-based, 0 = padding
    charges: np.ndarray               # (n_res, max_atom) float64
    is_active: np.ndarray             # (n_res,) int32 (input%is_active)
    epsilon: np.ndarray               # (n_atom_types, n_atom_types) float64, Kelvin
    sigma: np.ndarray                 # (n_atom_types, n_atom_types) float64, Angstrom
    names: List[str] = field(default_factory=list)

    def __post_init__(self):
        self.atoms_in_res = np.ascontiguousarray(self.atoms_in_res, dtype=np.int32)
        self.atom_types = np.ascontiguousarray(self.atom_types, dtype=np.int32)
        self.charges = np.ascontiguousarray(self.charges, dtype=np.float64)
        self.is_active = np.ascontiguousarray(self.is_active, dtype=np.int32)
        self.epsilon = np.ascontiguousarray(self.epsilon, dtype=np.float64)
        self.sigma = np.ascontiguousarray(self.sigma, dtype=np.float64)
        assert self.atom_types.shape == self.charges.shape
        assert self.atom_types.shape[0] == self.n_res
        assert self.epsilon.shape == self.sigma.shape == (self.n_atom_types, self.n_atom_types)

    @property
    def n_res(self) -> int:
        return int(self.atoms_in_res.shape[0])

    @property
    def max_atom(self) -> int:
        return int(self.atom_types.shape[1])

    @property
    def n_atom_types(self) -> int:
        return int(self.epsilon.shape[0])


@dataclass
class System:
    """A configuration: topology + box + Ewald inputs + molecule coordinates.

    ``com[t]`` has shape (n_mol_t, 3); ``offsets[t]`` has shape (n_mol_t, atoms_in_res[t], 3).
    ``box_matrix`` is ``box%matrix`` (columns are cell vectors for the distance routine,
    geometry_utils.f90:126-129).
    """

    topo: Topology
    box_matrix: np.ndarray            # (3, 3)
    bounds_lo: np.ndarray             # (3,)
    real_space_cutoff: float
    ewald_tolerance: float
    temperature: float
    com: List[np.ndarray]
    offsets: List[np.ndarray]
    label: str = ""

    def __post_init__(self):
        self.box_matrix = np.ascontiguousarray(self.box_matrix, dtype=np.float64)
        self.bounds_lo = np.ascontiguousarray(self.bounds_lo, dtype=np.float64)
        self.com = [np.ascontiguousarray(c, dtype=np.float64).reshape(-1, 3) for c in self.com]
        self.offsets = [
            np.ascontiguousarray(o, dtype=np.float64).reshape(-1, int(self.topo.atoms_in_res[t]), 3)
            for t, o in enumerate(self.offsets)
        ]
        for t in range(self.topo.n_res):
            assert self.com[t].shape[0] == self.offsets[t].shape[0]
            assert self.com[t].shape[0] <= NB_MAX_MOLECULE

    @property
    def n_mol(self) -> np.ndarray:
        return np.array([c.shape[0] for c in self.com], dtype=np.int32)

    @property
    def n_atoms(self) -> int:
        return int(sum(int(c.shape[0]) * int(self.topo.atoms_in_res[t]) for t, c in enumerate(self.com)))

    def sites(self, t: int, m: int) -> np.ndarray:
        """Absolute site coordinates com + offset of molecule (t, m), 0-based, shape (n1, 3).

        The reference forms exactly this sum before every use (geometry_utils.f90:379-382,
        ewald_phase.f90:398-399), so storing the rounded sum is bit-equivalent.
        """
        return self.com[t][m][None, :] + self.offsets[t][m]

    def all_sites(self, t: int) -> np.ndarray:
        return self.com[t][:, None, :] + self.offsets[t]

    def copy(self) -> "System":
        return System(self.topo, self.box_matrix.copy(), self.bounds_lo.copy(), self.real_space_cutoff,
                      self.ewald_tolerance, self.temperature, [c.copy() for c in self.com],
                      [o.copy() for o in self.offsets], self.label)

    def is_triclinic(self) -> bool:
        m = self.box_matrix
        off = np.array([m[0, 1], m[0, 2], m[1, 0], m[1, 2], m[2, 0], m[2, 1]])
        return bool(np.max(np.abs(off)) > ERROR_TOL)

"""MANIAC's input surface: the ``.maniac`` input, the LAMMPS ``.data`` topology and the ``.inc``
parameter file, read into a ``System`` for the energy engine.

SURVEY.md section 8(f) row 2 -- the data formats on the input side of the hot path -- so that the
reference's own fixture files (/root/reference/tests/readers/**) load unchanged, produce the same
state as the reference's front end, and fail on the same malformed files.  What each step follows:

  .maniac   ParseInputFile, SortResidues, ValidateAndRescaleMoveProbabilities
            (/root/reference/src/input_parser.f90:297-601, :603-672, :86-110)
  .data     ReadLMPHeaderInfo, ParseLAMMPSBox (readers_utils.f90:56-248), ReadLAMMPSMasses,
            ReadLAMMPSAtoms, SortAtomsByOriginalID, DetectResiduePattern, DetectMolecules,
            RepairActiveMolecules, TransformCoordinate (data_parser.f90:193-291, :552-671,
            :1055-1181, :1205-1287, :1297-1378, :1386-1511), RepairMolecule / ComputeCOM
            (readers_utils.f90:11-50, :262-328), ApplyPBC (geometry_utils.f90:167-220)
  .inc      ReadParameters, ApplyLorentzBerthelot (parameters_parser.f90:20-182)
  fugacity  ConvertFugacity (prepare_utils.f90:48-73)

Several quirks of the reference are reproduced on purpose, because they decide the numbers:
  * a top-level keyword must start in column 1 (the value is sliced at len(keyword), so a leading
    blank makes the read fail: "Error reading <keyword>");
  * ``primary%atom_masses`` is filled from the Masses section by running index, not by atom type,
    and the centre of mass of a molecule is computed with ONE mass for all its atoms (the last one
    matched), i.e. it is the plain centroid scaled by that mass;
  * site charges / types of a residue are those of the LAST molecule of that residue in the file;
  * epsilon / sigma are assigned by atom type, then unset pairs are filled by Lorentz-Berthelot.
Fatal conditions raise ``ManiacInputError`` carrying the reference's stop code where it has one.
Bonds / angles / dihedrals / impropers are parsed and matched per residue only for the data-file writer (the hot path
never uses them).
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import List

import numpy as np

from .system import A3_TO_M3, ATM_TO_PA, ERROR_TOL, KB_JK, KB_KCALMOL, NB_MAX_MOLECULE, System, Topology


class ManiacInputError(Exception):
    """The reference would AbortRun / error stop here."""

    def __init__(self, msg, code=1):
        super().__init__(msg)
        self.code = code


@dataclass
class Residue:
    name: str = ""
    is_active: int = -1
    fugacity_atm: float = -1.0
    types: List[int] = field(default_factory=list)
    names: List[str] = field(default_factory=list)
    nb_atoms: int = 0


@dataclass
class ManiacInput:
    nb_block: int = 0
    nb_step: int = 0
    temperature: float = 0.0
    seed: int = 0
    has_seed: bool = False
    ewald_tolerance: float = 0.0
    real_space_cutoff: float = 0.0
    translation_step: float = 0.0
    rotation_step_angle: float = 0.0
    recalibrate_moves: bool = False
    translation_proba: float = 0.0
    rotation_proba: float = 0.0
    insertion_deletion_proba: float = 0.0
    swap_proba: float = 0.0
    probabilities_rescaled: bool = False
    residues: List[Residue] = field(default_factory=list)

    def fugacity_per_A3(self):
        """ConvertFugacity: atm -> molecules per cubic Angstrom (active residues only)."""
        out = []
        for r in self.residues:
            if r.is_active == 1:
                if r.fugacity_atm <= 0.0:
                    raise ManiacInputError("Invalid fugacity for active residue with ID = " + r.name)
                out.append(r.fugacity_atm * ATM_TO_PA * A3_TO_M3 / (KB_JK * self.temperature))
            else:
                out.append(r.fugacity_atm)
        return out


# ---- Fortran list-directed reads of one value ---------------------------------------------------

def _first_token(text):
    toks = re.split(r"[ \t,]+", text.strip(" \t,"))
    return toks[0] if toks and toks[0] != "" else None


def _read_int(text):
    t = _first_token(text)
    if t is None or not re.fullmatch(r"[+-]?\d+", t):
        return None
    return int(t)


def _read_real(text):
    t = _first_token(text)
    if t is None:
        return None
    t2 = re.sub(r"[dD]", "e", t)
    if not re.fullmatch(r"[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?", t2):
        return None
    return float(t2)


def _read_logical(text):
    t = _first_token(text)
    if t is None:
        return None
    t = t.lower().lstrip(".")
    if t.startswith("t"):
        return True
    if t.startswith("f"):
        return False
    return None


# ---- .maniac ------------------------------------------------------------------------------------

_REAL_KEYS = {
    "temperature": ("temperature", "gt0"), "ewald_tolerance": ("ewald_tolerance", "gt0"),
    "real_space_cutoff": ("real_space_cutoff", "gt0"), "translation_step": ("translation_step", "gt0"),
    "rotation_step_angle": ("rotation_step_angle", "gt0"), "translation_proba": ("translation_proba", "p"),
    "rotation_proba": ("rotation_proba", "p"), "insertion_deletion_proba": ("insertion_deletion_proba", "p"),
    "swap_proba": ("swap_proba", "p"),
}


def read_maniac_input(path) -> ManiacInput:
    """ParseInputFile + SortResidues + ValidateAndRescaleMoveProbabilities."""
    try:
        lines = open(path).read().split("\n")
    except OSError as e:
        raise ManiacInputError(f"I/O error on file: {path}") from e
    inp = ManiacInput()
    seen = set()
    in_block = False
    cur = Residue()
    for line in lines:
        if line[:1] == "#" or line.strip() == "":
            continue
        keyword = _first_token(line)
        # the reference slices the value at len(keyword): right only when the keyword starts in column 1
        rest = line[len(keyword):]
        if keyword in ("nb_block", "nb_step", "seed"):
            v = _read_int(rest)
            if v is None:
                raise ManiacInputError("Error reading " + keyword)
            setattr(inp, keyword, v)
            seen.add(keyword)
            if keyword == "seed":
                inp.has_seed = True
        elif keyword in _REAL_KEYS:
            attr, rule = _REAL_KEYS[keyword]
            v = _read_real(rest)
            if v is None:
                raise ManiacInputError("Error reading " + keyword)
            if rule == "gt0" and v <= 0.0:
                raise ManiacInputError(f"Invalid {keyword}: must be > 0")
            if rule == "p" and (v < 0.0 or v > 1.0):
                raise ManiacInputError(f"Invalid {keyword}: must be in [0,1]")
            setattr(inp, attr, v)
            seen.add(keyword)
        elif keyword == "recalibrate_moves":
            v = _read_logical(rest)
            if v is None:
                raise ManiacInputError("Error reading recalibrate_moves")
            inp.recalibrate_moves = v
            seen.add(keyword)
        elif keyword == "begin_residue":
            in_block = True
            cur = Residue()
            continue
        elif keyword == "end_residue":
            in_block = False
            inp.residues.append(cur)
            continue
        if in_block:
            toks = line.split()
            if len(toks) < 2:
                raise ManiacInputError("Error reading residue line: " + line.strip())
            token, val = toks[0], toks[1]
            if token == "name":
                cur.name = val[:10]
            elif token == "state":
                if val == "actif":
                    cur.is_active = 1
                elif val == "inactif":
                    cur.is_active = 0
                else:
                    raise ManiacInputError("Unknown residue state")
            elif token == "fugacity":
                v = _read_real(" ".join(toks[1:]))
                if v is not None:
                    cur.fugacity_atm = v
            elif token == "nb-atoms":
                v = _read_int(" ".join(toks[1:]))
                if v is not None:
                    cur.nb_atoms = v
            elif token == "types":
                cur.types = []
                for t in toks[1:]:
                    if not re.fullmatch(r"[+-]?\d+", t):
                        break
                    cur.types.append(int(t))
            elif token == "names":
                cur.names = [t[:10] for t in toks[1:]]
    for r in inp.residues:
        if r.is_active == 1 and r.fugacity_atm < 0.0:
            raise ManiacInputError("Fugacity not provided or invalid for active residue: " + r.name)
    for req in ("nb_block", "nb_step", "temperature", "real_space_cutoff", "ewald_tolerance", "translation_step",
                "rotation_step_angle"):
        if req not in seen:
            raise ManiacInputError("Missing required parameter: " + req)
    total = inp.translation_proba + inp.rotation_proba + inp.insertion_deletion_proba + inp.swap_proba
    if total < ERROR_TOL:
        raise ManiacInputError("Invalid move probabilities: all enabled moves have zero probability")
    # SortResidues: stable insertion sort by the smallest atom type of each residue
    order = list(range(len(inp.residues)))
    keys = [min(r.types) if r.types else 0 for r in inp.residues]
    for i in range(1, len(order)):
        k = order[i]
        j = i - 1
        while j >= 0 and keys[order[j]] > keys[k]:
            order[j + 1] = order[j]
            j -= 1
        order[j + 1] = k
    inp.residues = [inp.residues[k] for k in order]
    # ValidateAndRescaleMoveProbabilities
    if abs(total - 1.0) > ERROR_TOL:
        inp.probabilities_rescaled = True
        scale = 1.0 / total
        inp.translation_proba *= scale
        inp.rotation_proba *= scale
        inp.insertion_deletion_proba *= scale
        inp.swap_proba *= scale
    return inp


# ---- LAMMPS .data -------------------------------------------------------------------------------

def _f_modulo(a, p):
    r = math.fmod(a, p)
    if r != 0.0 and ((r < 0.0) != (p < 0.0)):
        r += p
    return r


def _header_count(lines, word, exclude=None):
    for ln in lines:
        t = ln.strip()
        if not t or t[0] == "!":
            continue
        if word in t and (exclude is None or exclude not in t):
            v = _read_int(t)
            if v is not None:
                return v
    return 0


def _section_lines(lines, title, n, ncols, err_read, err_parse, what):
    """`n` data lines after the section header; mirrors the Read* loops of data_parser.f90."""
    start = None
    for i, ln in enumerate(lines):
        if ln.lstrip().startswith(title):
            start = i
            break
    if start is None:
        raise ManiacInputError(f"No {what} found in data file", err_read)
    i = start + 1
    if i < len(lines) and lines[i].strip() == "":
        i += 1
    rows = []
    for k in range(n):
        if i >= len(lines):
            raise ManiacInputError(f"Unexpected end of file at {what} line {k + 1}", err_read)
        toks = lines[i].split()
        i += 1
        if len(toks) < ncols:
            raise ManiacInputError(f"Failed to parse {what} line: '{lines[i - 1].strip()}'", err_parse)
        rows.append(toks)
    return rows


def read_lammps_data(path, inp: ManiacInput):
    """Everything ReadLMPData builds for the hot path: box, residue templates, com + offsets."""
    try:
        text = open(path).read()
    except OSError as e:
        raise ManiacInputError(f"Error opening file: {path}") from e
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines = lines[:-1]
    n_atoms = _header_count(lines, "atoms")
    n_types = _header_count(lines, "atom types")
    counts = {w: _header_count(lines, w, exclude=w[:-1] + " types") for w in ("bonds", "angles", "dihedrals", "impropers")}
    # ---- ParseLAMMPSBox
    lo = np.zeros(3); hi = np.zeros(3); tilt = np.zeros(3); triclinic = False
    for ln in lines:
        toks = ln.split()
        if len(toks) >= 4 and _read_real(toks[0]) is not None and _read_real(toks[1]) is not None:
            tag = toks[2] + " " + toks[3]
            for d, name in enumerate(("xlo xhi", "ylo yhi", "zlo zhi")):
                if tag == name:
                    lo[d] = _read_real(toks[0]); hi[d] = _read_real(toks[1])
        if len(toks) >= 6 and toks[3:6] == ["xy", "xz", "yz"] and all(_read_real(t) is not None for t in toks[:3]):
            tilt[:] = [_read_real(t) for t in toks[:3]]
            triclinic = True
    for d, name in enumerate("xyz"):
        if abs(lo[d]) < 1.0e-11 or abs(hi[d]) < 1.0e-11:
            raise ManiacInputError(f"ParseLAMMPSBox: {name}lo {name}hi not found in input file!", 1)
    lx, ly, lz = hi - lo
    matrix = np.array([[lx, 0.0, 0.0], [tilt[0], ly, 0.0], [tilt[1], tilt[2], lz]])   # readers_utils.f90:242-245
    # ---- ReadLAMMPSMasses
    masses = np.zeros(max(n_types, 1))
    found = 0
    for i, ln in enumerate(lines):
        if ln.strip() == "Masses":
            j = i + 2
            for _ in range(n_atoms):
                if j >= len(lines):
                    break
                t = lines[j].strip()
                j += 1
                if t == "" or t[0] == "!":
                    continue
                toks = t.split()
                if len(toks) < 2 or _read_int(toks[0]) is None or _read_real(toks[1]) is None:
                    break
                k = _read_int(toks[0])
                if k < 1 or k > n_atoms:
                    break
                if k <= n_types:
                    masses[k - 1] = _read_real(toks[1])
                found += 1
            break
    if found == 0:
        raise ManiacInputError("No masses found in data file", 12)
    if found != n_types:
        raise ManiacInputError("Number of masses found in data file differs from declared atom types", 13)
    # primary%atom_masses(res, k): filled by RUNNING index over (residue, type slot), data_parser.f90:266-277
    res_mass = []
    run = 0
    for r in inp.residues:
        row = []
        for _ in r.types:
            row.append(masses[run] if (run < n_atoms and run < n_types) else 0.0)
            run += 1
        res_mass.append(row)
    # ---- ReadLAMMPSAtoms
    rows = _section_lines(lines, "Atoms", n_atoms, 7, 14, 15, "atom")
    orig = np.zeros(n_atoms, dtype=np.int64); typ = np.zeros(n_atoms, dtype=np.int64)
    q = np.zeros(n_atoms); xyz = np.zeros((n_atoms, 3))
    for k, toks in enumerate(rows):
        vals_i = [_read_int(toks[0]), _read_int(toks[1]), _read_int(toks[2])]
        vals_f = [_read_real(t) for t in toks[3:7]]
        if None in vals_i or None in vals_f:
            raise ManiacInputError("Failed to parse atom line: '" + " ".join(toks) + "'", 15)
        if vals_i[2] < 1 or vals_i[2] > n_types:
            raise ManiacInputError(f"Invalid atom type {vals_i[2]} (max allowed: {n_types})", 16)
        orig[k], typ[k] = vals_i[0], vals_i[2]
        q[k] = vals_f[0]
        xyz[k] = vals_f[1:4]
    # bonded sections (ReadLAMMPSBonds / Angles / Dihedrals / Impropers, data_parser.f90:673-1053): id, type, atom ids;
    # a malformed section stops with codes 24 / 25 like the reference
    bonded = {}
    for title, ncols in (("Bonds", 4), ("Angles", 5), ("Dihedrals", 6), ("Impropers", 6)):
        n = counts[title.lower()]
        recs = []
        if n > 0:
            for toks in _section_lines(lines, title, n, ncols, 24, 25, title.lower()[:-1]):
                vals = [_read_int(t) for t in toks[:ncols]]
                if any(v is None for v in vals):
                    raise ManiacInputError(f"Failed to parse {title.lower()[:-1]} line", 25)
                recs.append(vals)
        bonded[title.lower()] = recs
    # ---- SortAtomsByOriginalID (stable insertion sort on the original id)
    order = np.argsort(orig, kind="stable")
    orig, typ, q, xyz = orig[order], typ[order], q[order], xyz[order]
    # ---- DetectResiduePattern
    n_res = len(inp.residues)
    pattern = [[0] * max(1, r.nb_atoms) for r in inp.residues]
    cpt = [0] * n_res
    for k in range(n_atoms):
        for i, r in enumerate(inp.residues):
            if int(typ[k]) in r.types:
                pattern[i][cpt[i]] = int(typ[k])
                cpt[i] += 1
                if cpt[i] >= r.nb_atoms:
                    cpt[i] = 0
                break
    # ---- DetectMolecules
    max_atom = max(r.nb_atoms for r in inp.residues)
    atom_types = np.zeros((n_res, max_atom), dtype=np.int32)
    charges = np.zeros((n_res, max_atom))
    n_mol = [0] * n_res
    for i, r in enumerate(inp.residues):
        k = 0
        while k < n_atoms:
            if int(typ[k]) == pattern[i][0]:
                if k + r.nb_atoms > n_atoms:
                    raise ManiacInputError("Not enough atoms left in box to complete residue type ")
                for j in range(r.nb_atoms):
                    atom_types[i, j] = typ[k]
                    charges[i, j] = q[k]
                    if r.is_active == 1 and int(typ[k]) != pattern[i][j]:
                        raise ManiacInputError("Issue with atom order in data file")
                    k += 1
                n_mol[i] += 1
            else:
                k += 1
        if n_mol[i] > NB_MAX_MOLECULE:
            raise ManiacInputError("The number of molecules exceeds the maximum allowed", 11)
    # box%atom_ids(i, j): original ids of the LAST molecule detected for residue i (data_parser.f90:1249), which is
    # what DetectBondPerResidue & co. (data_parser.f90:320-550) match the bonded records against
    atom_ids = np.zeros((n_res, max_atom), dtype=np.int64)
    for i, r in enumerate(inp.residues):
        k = 0
        while k < n_atoms:
            if int(typ[k]) == pattern[i][0] and k + r.nb_atoms <= n_atoms:
                atom_ids[i, : r.nb_atoms] = orig[k:k + r.nb_atoms]
                k += r.nb_atoms
            else:
                k += 1
    per_res = {}
    for key, nat in (("bonds", 2), ("angles", 3), ("dihedrals", 4), ("impropers", 4)):
        tables = []
        for i, r in enumerate(inp.residues):
            ids = [int(v) for v in atom_ids[i, : r.nb_atoms]]
            rows_i = []
            for rec in bonded[key]:
                members = rec[2:2 + nat]
                if all(a in ids for a in members):
                    loc = [ids.index(a) + 1 for a in members]          # atomIndexInResidue: first match, 1-based
                    if key == "bonds":
                        loc = sorted(loc)                               # (min, max), data_parser.f90:361-367
                    elif not (loc[0] < loc[-1]):
                        loc = loc[::-1]                                 # reversed unless first < last (:414-422, :472-482)
                    rows_i.append([rec[1]] + loc)
            tables.append(rows_i)
        per_res[key] = tables
    # box type and reciprocal as the reference computes them (geometry_utils.f90:68-154, :277-331)
    off_diag = [matrix[0, 1], matrix[0, 2], matrix[1, 0], matrix[1, 2], matrix[2, 0], matrix[2, 1]]
    if max(abs(v) for v in off_diag) > ERROR_TOL:
        box_type = 3
    elif abs(matrix[0, 0] - matrix[1, 1]) > ERROR_TOL or abs(matrix[0, 0] - matrix[2, 2]) > ERROR_TOL:
        box_type = 2
    else:
        box_type = 1
    a, b, c = matrix[:, 0], matrix[:, 1], matrix[:, 2]

    def cross(u, v):
        return np.array([u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]])
    adj = np.stack([cross(b, c), cross(c, a), cross(a, b)], axis=1)
    det = a[0] * adj[0, 0] + a[1] * adj[1, 0] + a[2] * adj[2, 0]
    if abs(det) < 1.0:
        raise ManiacInputError("Error: Determinant fell into denormal/underflow range")
    recip = (1.0 / det) * adj

    def wrap_nearest(x, boxlen):
        return _f_modulo(x + 0.5 * boxlen, boxlen) - 0.5 * boxlen

    # ---- RepairActiveMolecules (walks the atoms residue by residue in file order)
    xyz = xyz.copy()
    k = 0
    for i, r in enumerate(inp.residues):
        for _ in range(n_mol[i]):
            if r.is_active == 1:
                for ia in range(1, r.nb_atoms):
                    ref = xyz[k + ia - 1]
                    d = xyz[k + ia] - ref
                    if box_type in (1, 2):
                        d = np.array([wrap_nearest(d[dd], matrix[dd, dd]) for dd in range(3)])
                    else:
                        f = np.array([recip[ii, 0] * d[0] + recip[ii, 1] * d[1] + recip[ii, 2] * d[2] for ii in range(3)])
                        f = np.array([wrap_nearest(v, 1.0) for v in f])
                        d = np.array([matrix[ii, 0] * f[0] + matrix[ii, 1] * f[1] + matrix[ii, 2] * f[2] for ii in range(3)])
                    xyz[k + ia] = ref + d
            k += r.nb_atoms
    # ---- TransformCoordinate
    com_all, off_all = [], []
    for i, r in enumerate(inp.residues):
        mass = 0.0
        for j in range(r.nb_atoms):
            for l, ty in enumerate(r.types):
                if ty == int(atom_types[i, j]):
                    mass = res_mass[i][l]          # ONE mass for the whole molecule: the last match
        coms, offs = [], []
        k = 0
        while k < n_atoms:
            if n_mol[i] > 0 and int(typ[k]) == int(atom_types[i, 0]):
                mol = xyz[k:k + r.nb_atoms]
                if mol.shape[0] < r.nb_atoms:
                    break
                k += r.nb_atoms
                cx = cy = cz = tot = 0.0
                for p in mol:                       # ComputeCOM, readers_utils.f90:11-50
                    cx = cx + mass * p[0]; cy = cy + mass * p[1]; cz = cz + mass * p[2]
                    tot = tot + mass
                if tot <= 0.0:
                    raise ManiacInputError("Total mass is zero or negative", 1)
                com0 = np.array([cx / tot, cy / tot, cz / tot])
                com = com0.copy()
                if not triclinic:                   # ApplyPBC
                    for d in range(3):
                        com[d] = lo[d] + _f_modulo(com[d] - lo[d], matrix[d, d])
                else:
                    v = com - lo
                    f = [recip[ii, 0] * v[0] + recip[ii, 1] * v[1] + recip[ii, 2] * v[2] for ii in range(3)]
                    f = [_f_modulo(x, 1.0) for x in f]
                    com = np.array([lo[ii] + (matrix[ii, 0] * f[0] + matrix[ii, 1] * f[1] + matrix[ii, 2] * f[2])
                                    for ii in range(3)])
                coms.append(com)
                offs.append(mol - com0[None, :])
            else:
                k += 1
        com_all.append(np.array(coms).reshape(-1, 3))
        off_all.append(np.array(offs).reshape(-1, r.nb_atoms, 3))
    return dict(n_atoms=n_atoms, n_atom_types=n_types, matrix=matrix, lo=lo, hi=hi, triclinic=triclinic,
                box_type=box_type, atom_types=atom_types, charges=charges, n_mol=n_mol, com=com_all, off=off_all,
                masses=masses, tilt=tilt, atom_ids=atom_ids, bonded_per_residue=per_res,
                type_counts={w: _header_count(lines, w[:-1] + " types") for w in ("bonds", "angles", "dihedrals", "impropers")},
                bonded_counts=counts)


# ---- .inc ---------------------------------------------------------------------------------------

def read_parameters(path, n_atom_types, present_types, with_raw=False):
    """ReadParameters + ApplyLorentzBerthelot on an atom-type table (epsilon in K, sigma in A).

    ``present_types``: atom types that occur in some residue; the reference only ever stores
    coefficients for those."""
    try:
        lines = open(path).read().split("\n")
    except OSError as e:
        raise ManiacInputError(f"Error opening file: {path}") from e
    eps = np.zeros((n_atom_types, n_atom_types)); sig = np.zeros((n_atom_types, n_atom_types))
    for ln in lines:
        if ln[:1] == "#" or ln.strip() == "":
            continue
        toks = ln.split()
        if toks[0] != "pair_coeff":
            continue
        vals = toks[1:5]
        if len(vals) < 4 or _read_int(vals[0]) is None or _read_int(vals[1]) is None or \
                _read_real(vals[2]) is None or _read_real(vals[3]) is None:
            raise ManiacInputError("Failed to read pair_coeff value", 1)
        i, j = _read_int(vals[0]), _read_int(vals[1])
        e = _read_real(vals[2]) / KB_KCALMOL
        s = _read_real(vals[3])
        if 1 <= i <= n_atom_types and 1 <= j <= n_atom_types and i in present_types and j in present_types:
            eps[i - 1, j - 1] = eps[j - 1, i - 1] = e
            sig[i - 1, j - 1] = sig[j - 1, i - 1] = s
    filled_e, filled_s = eps.copy(), sig.copy()
    for i in present_types:
        for j in present_types:
            if abs(eps[i - 1, j - 1]) < 1.0e-6 and abs(sig[i - 1, j - 1]) < 1.0e-6:
                s = (sig[i - 1, i - 1] + sig[j - 1, j - 1]) / 2
                e = math.sqrt(eps[i - 1, i - 1] * eps[j - 1, j - 1])
                if s > 1.0e-6 and e > 1.0e-6:
                    filled_s[i - 1, j - 1] = s
                    filled_e[i - 1, j - 1] = e
    if with_raw:
        return filled_e, filled_s, eps, sig
    return filled_e, filled_s


# ---- everything ---------------------------------------------------------------------------------

def load_system(maniac_path, data_path, inc_path, with_data=False):
    """The reference's front end (main.f90:16-26) -> (System, ManiacInput) [, the parsed data-file dict]."""
    inp = read_maniac_input(maniac_path)
    dat = read_lammps_data(data_path, inp)
    present = sorted({int(t) for i, r in enumerate(inp.residues) for t in dat["atom_types"][i, : r.nb_atoms] if t > 0})
    eps, sig = read_parameters(inc_path, dat["n_atom_types"], present)
    topo = Topology(atoms_in_res=[r.nb_atoms for r in inp.residues], atom_types=dat["atom_types"],
                    charges=dat["charges"], is_active=[r.is_active for r in inp.residues], epsilon=eps, sigma=sig,
                    names=[r.name for r in inp.residues])
    # unused padding types must still be valid ids for the engine
    topo.atom_types[topo.atom_types == 0] = 0
    system = System(topo, dat["matrix"], dat["lo"], inp.real_space_cutoff, inp.ewald_tolerance, inp.temperature,
                    dat["com"], dat["off"], label="from_files")
    if with_data:
        return system, inp, dat
    return system, inp


# ---- log.maniac header --------------------------------------------------------------------------

MANIAC_VERSION = "v0.3.0-beta"      # the reference version this front end mirrors (its version_module string)


def _box78(text):
    return "| " + text.ljust(74)[:74] + " |"


def _warn(msg):
    """WarnUser, output_utils.f90:569-585."""
    bar = "-" * 50
    return [bar, "WARNING:", msg, "Execution will continue.", bar]


def _f(value, w, d):
    """Fortran Fw.d edit descriptor (asterisks on overflow)."""
    t = f"{value:{w}.{d}f}"
    return t if len(t) <= w else "*" * w


def _data_block(name, dat, inp, is_primary, masses_n):
    """PrepareSimulationBox + the section readers' INFO lines + LogData + LogConnectivity for one data file
    (geometry_utils.f90:25-56, data_parser.f90:690-990, output_utils.f90:352-470)."""
    from .engine import box_prepare
    box_type, volume, _, _ = box_prepare(dat["matrix"])
    out = ["====== Simulation preparation ======", "",
           {1: "Box symmetry type: Cubic", 2: "Box symmetry type: Orthorhombic", 3: "Box symmetry type: Triclinic"}.get(
               box_type, f"Box symmetry type determined: {box_type}"),
           "Cell volume (Å^3): " + _f(volume, 20, 4)]
    for key in ("bonds", "angles", "dihedrals", "impropers"):
        if dat["bonded_counts"][key] == 0:
            out.append(f"INFO: No {key} expected in data file: {name}")
    out += ["", "====== Import data file ======", f"Reading file {name}", "",
            f"Number of atoms: {dat['n_atoms']}", f"Number of type of residues: {len(inp.residues)}",
            f"Number of type of atoms: {dat['n_atom_types']}"]
    last = None
    for i, r in enumerate(inp.residues):
        n = int(dat["n_mol"][i])
        if n != 0 and r.is_active == 1:
            last = f"Active residue {r.name} found in the data file: {n}"
        elif n != 0 and r.is_active == 0:
            last = f"Inactive residue {r.name} found in the data file: {n}"
        # a residue absent from the file repeats the previous message (LogData logs formatted_msg unconditionally)
        out.append(last if last is not None else "")
    m = np.asarray(dat["matrix"])
    out += ["", "Simulation box (rows):"] + ["".join(_f(m[i, j], 12, 6) for j in range(3)) for i in range(3)]
    out += ["", "Atoms masses (g/mol):"] + [f"{k + 1:5d}  " + _f(dat["masses"][k], 12, 6) for k in range(masses_n)]
    if dat["bonded_counts"]["bonds"] > 0 or dat["bonded_counts"]["angles"] > 0:
        out += ["", "===== Connectivity summary =====", ""]
        for kind, word, ncol in (("bonds", "bond", 3), ("angles", "angle", 4)):
            for i, r in enumerate(inp.residues):
                if int(dat["n_mol"][i]) > 0:
                    rows = dat["bonded_per_residue"][kind][i]
                    out.append(f"Residue {r.name}: {len(rows)} {kind}")
                    for row in rows[:6]:
                        out.append(f"   {word} type {row[0]}: atoms [" + ",".join(str(v) for v in row[1:ncol]) + "]")
                    if len(rows) > 6:
                        out.append(f"   ... {len(rows) - 6} more {kind} not shown")
            if kind == "bonds":
                out.append("")
    return out


def log_header_lines(inp: ManiacInput, dat, maniac_name, data_name, inc_name, eps_raw, sig_raw, ewald,
                     reservoir=None):
    """The messages the reference logs before "Started Monte Carlo Loop", in its order: banner (WriteHeader,
    initoutput_utils.f90:68-82), input echo (PrintInputSummary, output_utils.f90:655-735), the data-file blocks (primary,
    then the reservoir), the Lorentz-Berthelot listing (parameters_parser.f90:116-182), LogParameters and
    LogEwaldParameters (prepare_utils.f90:75-97).  One string per LogMessage call; the Fortran side writes each with
    the reference's list-directed write, so line wrapping is the runtime's own.

    ``eps_raw`` / ``sig_raw``: the pair table as read (K, Angstrom), before the Lorentz-Berthelot fill;
    ``ewald``: dict(rc, tol, screening, alpha, fourier_precision, kmax, nk); ``reservoir``: (file name, parsed dict)."""
    rule = "+" + "-" * 76 + "+"
    out = ["", rule, _box78("MANIAC-MC - Version " + MANIAC_VERSION),
           _box78("Code written and maintained by Simon Gravelle, LIPhy, CNRS"), rule, ""]
    if inp.probabilities_rescaled:
        out += _warn("Move probabilities rescaled to sum to 1.0")
    out += ["====== Import input file ======", "", f"Reading file {maniac_name}", "", "=== Generic parameters",
            f"Number of blocks: {inp.nb_block}", f"Number of steps: {inp.nb_step}",
            "Temperature (K): " + _f(inp.temperature, 10, 2), "", "=== Electrostatic interactions",
            "Ewald tolerance: " + _f(inp.ewald_tolerance, 15, 8), "Cutoff (Å): " + _f(inp.real_space_cutoff, 10, 2), "",
            "=== Monte carlo move", "Translation step (Å): " + _f(inp.translation_step, 10, 2),
            "Rotation step angle (radian): " + _f(inp.rotation_step_angle, 10, 2),
            "Translation proba: " + _f(inp.translation_proba, 10, 2), "Rotation proba: " + _f(inp.rotation_proba, 10, 2),
            "Insertion deletion proba: " + _f(inp.insertion_deletion_proba, 10, 2), "Swap proba: " + _f(inp.swap_proba, 10, 2),
            "", "=== Residue information", "", f"Number of type of residue found: {len(inp.residues)}", ""]
    for r in inp.residues:
        out += [f"  Residue {r.name}", "  Is active: " + ("yes" if r.is_active == 1 else "no")]
        if r.is_active == 1:
            out.append("  Fugacity (atm): " + _f(r.fugacity_atm, 10, 2))
        out += [f"  Number of atoms in residue: {r.nb_atoms}", f"  Number of atom types in residue: {len(r.types)}",
                "  Types:" + "".join(f" {t}" for t in r.types), "  Names:" + "".join(f" {n}" for n in r.names), ""]
    out += _data_block(data_name, dat, inp, True, dat["n_atom_types"])
    if reservoir is not None:
        out += _data_block(reservoir[0], reservoir[1], inp, False, dat["n_atom_types"])
    # ApplyLorentzBerthelot: pairs in the order the 4-deep loop (residue, atom, residue, atom) first meets them
    warned = set()
    types = dat["atom_types"]
    lb = []
    for i, ri in enumerate(inp.residues):
        for k in range(ri.nb_atoms):
            for j, rj in enumerate(inp.residues):
                for l in range(rj.nb_atoms):
                    ti, tj = int(types[i, k]), int(types[j, l])
                    if ti <= 0 or tj <= 0:
                        continue
                    if abs(eps_raw[ti - 1, tj - 1]) < 1.0e-6 and abs(sig_raw[ti - 1, tj - 1]) < 1.0e-6:
                        sg = (sig_raw[ti - 1, ti - 1] + sig_raw[tj - 1, tj - 1]) / 2
                        ep = math.sqrt(eps_raw[ti - 1, ti - 1] * eps_raw[tj - 1, tj - 1])
                        if sg > 1.0e-6 and ep > 1.0e-6 and (ti, tj) not in warned:
                            if not warned:
                                lb += ["INFO: Enforcing the Lorentz-Berthelot rule", "typei typej epsilon sigma"]
                            a, b = (ti, tj) if ti < tj else (tj, ti)
                            lb.append(f"{a:3d} -{b:3d} : " + _f(sg, 8, 4) + " Å " + _f(ep * KB_KCALMOL, 8, 4) + " kcal/mol")
                            warned.add((ti, tj)); warned.add((tj, ti))
    out += lb
    km = ewald["kmax"]
    out += ["", "====== Import parameter file ======", "", f"Reading file {inc_name}",
            "Real-space cutoff (Å): " + _f(ewald["rc"], 10, 4),
            "Ewald accuracy tolerance: " + "%12s" % ("%.5E" % ewald["tol"]),
            "Screening factor (dimensionless): " + _f(ewald["screening"], 10, 4),
            "Ewald damping parameter alpha (1/Å): " + _f(ewald["alpha"], 10, 4),
            "Fourier-space precision parameter: " + _f(ewald["fourier_precision"], 10, 4),
            f"Max Fourier index (kmax(1), kmax(2), kmax(3)): {int(km[0]):5d}, {int(km[1]):5d}, {int(km[2]):5d}",
            f"Total reciprocal lattice vectors: {int(ewald['nk']):10d}"]
    return out


# ---- writing input files ------------------------------------------------------------------------

def write_input_files(system: System, directory, *, nb_block, nb_step, translation_step, rotation_step_angle,
                      translation_proba, rotation_proba, insertion_deletion_proba=0.0, fugacity_atm=None,
                      recalibrate_moves=False, seed=None, atom_names=None, masses=None, stem="system"):
    """Write `system` as the three files the reference reads (`.maniac`, LAMMPS `.data` atom_style full,
    `.inc`); returns their paths.  Coordinates are printed with 17 significant digits, epsilon in
    kcal/mol (the reference divides by KB_kcalmol when it reads them, parameters_parser.f90:89-98).
    Residue types must already be ordered by their smallest atom type (SortResidues would reorder them).
    """
    import os
    topo = system.topo
    os.makedirs(directory, exist_ok=True)
    n_res = topo.n_res
    res_types = [[int(t) for t in topo.atom_types[i, : topo.atoms_in_res[i]]] for i in range(n_res)]
    assert [min(r) for r in res_types] == sorted(min(r) for r in res_types), "order residues by smallest atom type"
    names = list(topo.names) if topo.names else [f"res{i + 1}" for i in range(n_res)]
    if fugacity_atm is None:
        fugacity_atm = [1.0] * n_res
    nt = topo.n_atom_types
    if masses is None:
        masses = [1.0] * nt
    if atom_names is None:
        atom_names = [f"A{t + 1}" for t in range(nt)]
    p_maniac = os.path.join(directory, stem + ".maniac")
    p_data = os.path.join(directory, stem + ".data")
    p_inc = os.path.join(directory, stem + ".inc")
    with open(p_maniac, "w") as f:
        f.write("# generated input\n")
        f.write(f"nb_block {int(nb_block)}\nnb_step {int(nb_step)}\ntemperature {float(system.temperature)!r}\n")
        if seed is not None:
            f.write(f"seed {int(seed)}\n")
        f.write(f"ewald_tolerance {float(system.ewald_tolerance)!r}\nreal_space_cutoff {float(system.real_space_cutoff)!r}\n")
        f.write(f"translation_step {float(translation_step)!r}\nrotation_step_angle {float(rotation_step_angle)!r}\n")
        f.write(f"recalibrate_moves {'true' if recalibrate_moves else 'false'}\n")
        f.write(f"translation_proba {float(translation_proba)!r}\nrotation_proba {float(rotation_proba)!r}\n")
        f.write(f"insertion_deletion_proba {float(insertion_deletion_proba)!r}\n\n")
        for i in range(n_res):
            uniq = []
            for t in res_types[i]:
                if t not in uniq:
                    uniq.append(t)
            f.write("begin_residue\n")
            f.write(f"  name {names[i]}\n  state {'actif' if topo.is_active[i] == 1 else 'inactif'}\n")
            if topo.is_active[i] == 1:
                f.write(f"  fugacity {float(fugacity_atm[i])!r}\n")
            f.write("  types " + " ".join(str(t) for t in uniq) + "\n")
            f.write("  names " + " ".join(atom_names[t - 1] for t in uniq) + "\n")
            f.write(f"  nb-atoms {int(topo.atoms_in_res[i])}\n")
            f.write("end_residue\n\n")
    hi = system.bounds_lo + np.diag(system.box_matrix)
    with open(p_data, "w") as f:
        f.write("LAMMPS data file (atom_style full), generated\n\n")
        f.write(f"{system.n_atoms} atoms\n{nt} atom types\n0 bonds\n0 bond types\n0 angles\n0 angle types\n")
        f.write("0 dihedrals\n0 dihedral types\n0 impropers\n0 improper types\n\n")
        for d, ax in enumerate("xyz"):
            f.write(f"{float(system.bounds_lo[d])!r} {float(hi[d])!r} {ax}lo {ax}hi\n")
        if system.is_triclinic():
            m = system.box_matrix
            f.write(f"{float(m[1, 0])!r} {float(m[2, 0])!r} {float(m[2, 1])!r} xy xz yz\n")
        f.write("\nMasses\n\n")
        for t in range(nt):
            f.write(f"{t + 1} {float(masses[t])!r}\n")
        f.write("\nAtoms\n\n")
        atom_id = mol_id = 0
        for i in range(n_res):
            sites = system.all_sites(i)
            for m in range(sites.shape[0]):
                mol_id += 1
                for a in range(int(topo.atoms_in_res[i])):
                    atom_id += 1
                    x, y, z = (float(v) for v in sites[m, a])
                    f.write(f"{atom_id} {mol_id} {res_types[i][a]} {float(topo.charges[i, a])!r} {x!r} {y!r} {z!r}\n")
    with open(p_inc, "w") as f:
        for t in range(nt):
            f.write(f"pair_coeff {t + 1} {t + 1} {float(topo.epsilon[t, t] * KB_KCALMOL)!r} {float(topo.sigma[t, t])!r}\n")
    return p_maniac, p_data, p_inc

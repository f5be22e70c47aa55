"""OpenFOAM case reader (rapidcfd-dev_amd/foam/polyMesh.{H,C}) + polyMeshFoam: the test writes a constant/polyMesh in
OpenFOAM's on-disk format (ascii and binary), recomputes geometry and Laplacian coefficients with numpy and checks the
application's output -- the CPU test runs the reader/geometry only (no device), the gpu test the whole solve."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rapidcfd-dev_amd")

HEADER = """/*--------------------------------*- C++ -*----------------------------------*\\
| =========                 |                                                 |
\\*---------------------------------------------------------------------------*/
FoamFile
{{
    version     2.0;
    format      {fmt};
    class       {cls};
    note        "{note}";
    location    "constant/polyMesh";
    object      {obj};
}}
// * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * //

"""


def make_box_mesh(dims, seed=5):
    """points / faces / owner / neighbour / patches of a distorted hex box in OpenFOAM's ordering"""
    nx, ny, nz = dims
    gx, gy, gz = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), np.arange(nz + 1), indexing="ij")
    x, y, z = gx / nx, gy / nx, gz / nx
    rng = np.random.default_rng(seed)
    # smooth grading + a little vertex noise (non-planar faces, non-orthogonality)
    X = x + 0.08 * np.sin(2 * y + 1.0) * x * (1 - x) + 0.1 / nx * (rng.random(x.shape) - 0.5) * (gx > 0) * (gx < nx)
    Y = y + 0.05 * np.sin(3 * x) * y * (ny / nx - y) + 0.1 / nx * (rng.random(x.shape) - 0.5) * (gy > 0) * (gy < ny)
    Z = z * (1 + 0.2 * x) + 0.1 / nx * (rng.random(x.shape) - 0.5) * (gz > 0) * (gz < nz)
    if seed is None:                       # undistorted box (the periodic self-coupling test needs matching faces)
        X, Y, Z = x, y, z
    pid = lambda i, j, k: i + (nx + 1) * (j + (ny + 1) * k)
    pts = np.zeros(((nx + 1) * (ny + 1) * (nz + 1), 3))
    pts[pid(gx, gy, gz).ravel()] = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    cid = lambda i, j, k: i + nx * (j + ny * k)

    def face(i, j, k, d, outward_plus):   # quad of cell (i,j,k) on its +d side, normal along +d
        if d == 0:
            q = [pid(i + 1, j, k), pid(i + 1, j + 1, k), pid(i + 1, j + 1, k + 1), pid(i + 1, j, k + 1)]
        elif d == 1:
            q = [pid(i, j + 1, k), pid(i, j + 1, k + 1), pid(i + 1, j + 1, k + 1), pid(i + 1, j + 1, k)]
        else:
            q = [pid(i, j, k + 1), pid(i + 1, j, k + 1), pid(i + 1, j + 1, k + 1), pid(i, j + 1, k + 1)]
        return q if outward_plus else q[::-1]

    faces, owner, neighbour = [], [], []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                for d, ok, nb in ((0, i < nx - 1, cid(i + 1, j, k)), (1, j < ny - 1, cid(i, j + 1, k)), (2, k < nz - 1, cid(i, j, k + 1))):
                    if ok:
                        faces.append(face(i, j, k, d, True)); owner.append(cid(i, j, k)); neighbour.append(nb)
    patches = []

    def add_patch(name, ptype, items):
        start = len(faces)
        for f, c in items:
            faces.append(f); owner.append(c)
        patches.append((name, ptype, len(items), start))
    add_patch("inlet", "patch", [(face(-1, j, k, 0, False), cid(0, j, k)) for k in range(nz) for j in range(ny)])
    add_patch("outlet", "patch", [(face(nx - 1, j, k, 0, True), cid(nx - 1, j, k)) for k in range(nz) for j in range(ny)])
    walls = [(face(i, -1, k, 1, False), cid(i, 0, k)) for k in range(nz) for i in range(nx)]
    walls += [(face(i, ny - 1, k, 1, True), cid(i, ny - 1, k)) for k in range(nz) for i in range(nx)]
    walls += [(face(i, j, -1, 2, False), cid(i, j, 0)) for j in range(ny) for i in range(nx)]
    walls += [(face(i, j, nz - 1, 2, True), cid(i, j, nz - 1)) for j in range(ny) for i in range(nx)]
    add_patch("walls", "wall", walls)
    return pts, np.array(faces, dtype=np.int32), np.array(owner, dtype=np.int32), np.array(neighbour, dtype=np.int32), patches


def write_case(case_dir, pts, faces, owner, neighbour, patches, source, binary):
    pm = os.path.join(case_dir, "constant", "polyMesh")
    os.makedirs(pm, exist_ok=True)
    os.makedirs(os.path.join(case_dir, "0"), exist_ok=True)
    fmt = "binary" if binary else "ascii"
    ncells = int(owner.max()) + 1
    note = f"nPoints:{len(pts)}  nCells:{ncells}  nFaces:{len(faces)}  nInternalFaces:{len(neighbour)}"

    def put(name, cls, body_ascii, body_binary):
        with open(os.path.join(pm, name), "wb") as f:
            f.write(HEADER.format(fmt=fmt, cls=cls, note=note, obj=name).encode())
            f.write(body_binary() if binary else body_ascii().encode())
            f.write(b"\n\n// ************************************************************************* //\n")

    def labels_bin(a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        return f"{len(a)}(".encode() + a.tobytes() + b")"
    put("points", "vectorField",
        lambda: f"{len(pts)}\n(\n" + "\n".join(f"({p[0]!r} {p[1]!r} {p[2]!r})" for p in pts.tolist()) + "\n)\n",
        lambda: f"{len(pts)}(".encode() + np.ascontiguousarray(pts, dtype=np.float64).tobytes() + b")")
    put("faces", "faceCompactList" if binary else "faceList",
        lambda: f"{len(faces)}\n(\n" + "\n".join("4(" + " ".join(map(str, f)) + ")" for f in faces.tolist()) + "\n)\n",
        lambda: labels_bin(np.arange(0, 4 * len(faces) + 1, 4)) + b"\n" + labels_bin(faces.ravel()))
    put("owner", "labelList", lambda: f"{len(owner)}\n(\n" + "\n".join(map(str, owner.tolist())) + "\n)\n", lambda: labels_bin(owner))
    put("neighbour", "labelList", lambda: f"{len(neighbour)}\n(\n" + "\n".join(map(str, neighbour.tolist())) + "\n)\n", lambda: labels_bin(neighbour))
    with open(os.path.join(pm, "boundary"), "w") as f:
        f.write(HEADER.format(fmt="ascii", cls="polyBoundaryMesh", note=note, obj="boundary"))
        f.write(f"{len(patches)}\n(\n")
        for pt in patches:
            name, ptype, n, start = pt[:4]
            f.write(f"    {name}\n    {{\n        type            {ptype};\n" + ("        inGroups        1(wall);\n" if ptype == "wall" else "")
                    + f"        nFaces          {n};\n        startFace       {start};\n"
                    + (f"        matchTolerance  0.0001;\n        myProcNo        {pt[4]};\n        neighbProcNo    {pt[5]};\n" if ptype == "processor" else "") + "    }\n")
        f.write(")\n")
    with open(os.path.join(case_dir, "0", "S"), "w") as f:
        f.write(HEADER.format(fmt="ascii", cls="volScalarField", note="", obj="S").replace('location    "constant/polyMesh"', 'location    "0"'))
        f.write("dimensions      [0 0 -1 0 0 0 0];\n\ninternalField   nonuniform List<scalar> \n" + f"{len(source)}\n(\n" + "\n".join(repr(v) for v in source.tolist())
                + "\n)\n;\n\nboundaryField\n{\n    inlet { type fixedValue; value uniform 0; }\n    outlet { type fixedValue; value uniform 0; }\n    walls { type zeroGradient; }\n}\n")


def geometry(pts, faces, owner, neighbour):
    """numpy restatement of primitiveMeshFaceCentresAndAreas.C / CellCentresAndVols.C / surfaceInterpolation.C"""
    P = pts[faces]                                    # [F, 4, 3]
    fc = P.sum(axis=1) / 4.0
    sumN = np.zeros((len(faces), 3)); sumA = np.zeros(len(faces)); sumAc = np.zeros((len(faces), 3))
    for pi in range(4):
        p0, p1 = P[:, pi], P[:, (pi + 1) % 4]
        c = p0 + p1 + fc
        nrm = np.cross(p1 - p0, fc - p0)
        a = np.sqrt((nrm * nrm).sum(axis=1))
        sumN += nrm; sumA += a; sumAc += a[:, None] * c
    Cf = ((1.0 / 3.0) / sumA)[:, None] * sumAc
    Sf = 0.5 * sumN
    magSf = np.sqrt((Sf * Sf).sum(axis=1))
    n = int(owner.max()) + 1
    nI = len(neighbour)
    cEst = np.zeros((n, 3)); cnt = np.zeros(n)
    np.add.at(cEst, owner, Cf); np.add.at(cnt, owner, 1)
    np.add.at(cEst, neighbour, Cf[:nI]); np.add.at(cnt, neighbour, 1)
    cEst /= cnt[:, None]
    C = np.zeros((n, 3)); V = np.zeros(n)
    pyr = (Sf * (Cf - cEst[owner])).sum(axis=1)
    np.add.at(C, owner, pyr[:, None] * (0.75 * Cf + 0.25 * cEst[owner])); np.add.at(V, owner, pyr)
    pyr = (Sf[:nI] * (cEst[neighbour] - Cf[:nI])).sum(axis=1)
    np.add.at(C, neighbour, pyr[:, None] * (0.75 * Cf[:nI] + 0.25 * cEst[neighbour])); np.add.at(V, neighbour, pyr)
    C /= V[:, None]; V /= 3.0
    sOwn = np.abs((Sf[:nI] * (Cf[:nI] - C[owner[:nI]])).sum(axis=1)); sNei = np.abs((Sf[:nI] * (C[neighbour] - Cf[:nI])).sum(axis=1))
    w = sNei / (sOwn + sNei)
    d = C[neighbour] - C[owner[:nI]]
    nhat = Sf[:nI] / magSf[:nI, None]
    delta = 1.0 / np.maximum((nhat * d).sum(axis=1), 0.05 * np.sqrt((d * d).sum(axis=1)))
    db = Cf[nI:] - C[owner[nI:]]
    nb = Sf[nI:] / magSf[nI:, None]
    delta_b = 1.0 / np.maximum((nb * db).sum(axis=1), 0.05 * np.sqrt((db * db).sum(axis=1)))
    return dict(Cf=Cf, Sf=Sf, magSf=magSf, C=C, V=V, weights=w, delta=delta, delta_b=delta_b)


GEOM = re.compile(r"geometry: sumV (\S+) sumMagSfInternal (\S+) sumWeights (\S+) sumNonOrthDeltaCoeffs (\S+)")
LINE = re.compile(r"^(\w+):  Solving for (\w+), Initial residual = (\S+), Final residual = (\S+), No Iterations (\d+)")


@pytest.mark.gpu
@pytest.mark.parametrize("binary", [False, True])
def test_polyMeshFoam_reads_a_case_and_matches_the_oracle(pkg, orc, tmp_path, binary):
    dims = (12, 9, 7)
    pts, faces, owner, neighbour, patches = make_box_mesh(dims)
    n = int(owner.max()) + 1
    G = geometry(pts, faces, owner, neighbour)
    S = np.sin(4 * G["C"][:, 0]) * np.cos(3 * G["C"][:, 1]) + G["C"][:, 2]
    case_dir = str(tmp_path / "case")
    write_case(case_dir, pts, faces, owner, neighbour, patches, S, binary)
    out = subprocess.run([os.path.join(PKG, "polyMeshFoam"), case_dir, "-nonOrthCorrectors", "3", "-write", "1", "-writeFormat", "binary" if binary else "ascii",
                          "-writePrecision", "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert f"nCells {n} nFaces {len(faces)} nInternalFaces {len(neighbour)}" in out.stdout
    m = GEOM.search(out.stdout)
    nI = len(neighbour)
    for got, ref in zip(map(float, m.groups()), (G["V"].sum(), G["magSf"][:nI].sum(), G["weights"].sum(), G["delta"].sum())):
        assert abs(got - ref) < 1e-12 * abs(ref)
    assert abs(G["V"].sum() - np.linalg.det(np.eye(3))) < 1.0   # sanity: a box of order-one volume
    # the same Laplacian with numpy + the oracle
    syn = pkg.synthetic
    lo, up = owner[:nI], neighbour
    upper, diag = orc.fvm_laplacian(n, lo, up, G["delta"], G["magSf"][:nI])
    for name, ptype, cnt, start in patches:
        if ptype == "patch":
            fc = owner[start:start + cnt]
            ic = -(G["magSf"][start:start + cnt] * G["delta_b"][start - nI:start - nI + cnt])
            diag = orc.patch_add(fc, ic, diag, 0)
    src = S * G["V"]
    case = syn.LduCase(n, lo, up, diag, upper, None, src)
    z = np.zeros(n)
    _, p1 = orc.System([case]).pcg(z, src, "AINV", tolerance=1e-9)
    w = np.sqrt(((G["Sf"][:nI] / np.sqrt(G["magSf"][:nI])[:, None] * np.array([1.0, 1.01, 1.02])) ** 2).sum(axis=1))
    x2, p2 = orc.GamgHierarchy(case, w, 10).solve(z, src, tolerance=1e-9)
    got = [(mm.group(1), float(mm.group(3)), float(mm.group(4)), int(mm.group(5))) for mm in map(LINE.match, out.stdout.splitlines()) if mm]
    assert [g[0] for g in got] == ["AINVPCG", "GAMG"] + ["AINVPCG"] * 3
    # the non-orthogonal corrector loop (gaussLaplacianSchemes.C:64-90 + correctedSnGrad.C:45-65): the same sweeps on the oracle
    Sf = [np.ascontiguousarray(G["Sf"][:nI, k]) for k in range(3)]
    nhat = G["Sf"][:nI] / G["magSf"][:nI, None]
    cv = nhat - (G["C"][up] - G["C"][lo]) * G["delta"][:, None]
    cv = [np.ascontiguousarray(cv[:, k]) for k in range(3)]
    S_sys = orc.System([case])
    ph, corrected = np.zeros(n), []
    for _ in range(3):
        ssf = orc.face_interpolate(lo, up, G["weights"], ph)
        g3 = orc.gauss_grad(n, lo, up, Sf, ssf, None)
        for name, ptype, cnt, start in patches:
            fc = owner[start:start + cnt]
            bv = np.zeros(cnt) if ptype == "patch" else ph[fc]
            for k in range(3):
                g3[k] = orc.patch_add_product(fc, np.ascontiguousarray(G["Sf"][start:start + cnt, k]), bv, g3[k], 0)
        g3 = [x / G["V"] for x in g3]
        flux = orc.sngrad_correction_flux(lo, up, cv, G["weights"], g3, G["magSf"][:nI])
        srcc = orc.submul(G["V"], orc.surface_integrate(n, lo, up, flux, G["V"]), src)
        ph, pc = S_sys.pcg(ph, srcc, "AINV", tolerance=1e-10)
        corrected.append(pc)
    for g, p in zip(got[2:], corrected):
        assert g[3] == p["nIterations"], (g, p["nIterations"])
        assert abs(g[1] - p["initialResidual"]) < 1e-9 * max(p["initialResidual"], 1e-30) + 1e-12 and abs(g[2] - p["finalResidual"]) < 1e-8 * max(p["initialResidual"], 1e-30) + 1e-12
    assert corrected[1]["initialResidual"] < 0.5 * corrected[0]["initialResidual"]        # the correctors converge
    mc = re.search(r"corrected p sum max: (\S+) (\S+)", out.stdout)
    assert abs(float(mc.group(1)) - ph.sum()) < 1e-6 * np.abs(ph).sum() and abs(float(mc.group(2)) - np.abs(ph).max()) < 1e-6 * np.abs(ph).max()
    for g, p in zip(got[:2], (p1, p2)):
        assert g[3] == p["nIterations"], (g, p["nIterations"])
        assert abs(g[1] - p["initialResidual"]) < 1e-10 and abs(g[2] - p["finalResidual"]) < 1e-9 * max(p["initialResidual"], 1e-30) + 1e-10
    mm = re.search(r"p sum max: (\S+) (\S+)", out.stdout)
    assert abs(float(mm.group(1)) - x2.sum()) < 1e-6 * np.abs(x2).sum() and abs(float(mm.group(2)) - np.abs(x2).max()) < 1e-6 * np.abs(x2).max()
    assert out.stdout.strip().endswith("End")
    # the solution went back into the case as <case>/1/p: a volScalarField file as GeometricField writes it
    fld = read_vol_field(os.path.join(case_dir, "1", "p"))
    assert fld["header"]["class"] == "volScalarField" and fld["header"]["object"] == "p" and fld["header"]["location"] == "1"
    assert fld["header"]["format"] == ("binary" if binary else "ascii") and fld["dimensions"] == "[0 2 -2 0 0 0 0]"
    assert fld["internalField"].shape == (n,) and np.max(np.abs(fld["internalField"] - x2)) < 1e-7 * np.abs(x2).max()
    assert [b[0] for b in fld["boundaryField"]] == [pt[0] for pt in patches]
    for (name, entries), (_, ptype, cnt, _) in zip(fld["boundaryField"], patches):
        assert entries["type"] == ("fixedValue" if ptype == "patch" else "zeroGradient")
        assert ("value" in entries) == (ptype == "patch") and (ptype != "patch" or entries["value"] == ("uniform", 0.0))


def read_vol_field(path):
    """a volScalarField / volVectorField file, parsed independently of the C++ reader: FoamFile header entries, dimensions, internalField
    (uniform -> ("uniform", v); nonuniform -> numpy array; ascii or binary lists), boundaryField as [(patch, {key: value})]"""
    raw = open(path, "rb").read()
    pos = [0]

    def skip():
        while True:
            while pos[0] < len(raw) and raw[pos[0]:pos[0] + 1].isspace():
                pos[0] += 1
            if raw[pos[0]:pos[0] + 2] == b"//":
                pos[0] = raw.index(b"\n", pos[0])
                continue
            if raw[pos[0]:pos[0] + 2] == b"/*":
                pos[0] = raw.index(b"*/", pos[0]) + 2
                continue
            return

    def tok():
        skip()
        c = raw[pos[0]:pos[0] + 1]
        if c in b"(){};" and c:
            pos[0] += 1
            return c.decode()
        if c == b'"':
            e = raw.index(b'"', pos[0] + 1)
            t = raw[pos[0] + 1:e].decode(); pos[0] = e + 1
            return t
        b0 = pos[0]
        while pos[0] < len(raw) and not raw[pos[0]:pos[0] + 1].isspace() and raw[pos[0]:pos[0] + 1] not in b"(){};":
            pos[0] += 1
        return raw[b0:pos[0]].decode()

    assert tok() == "FoamFile" and tok() == "{"
    header = {}
    while True:
        k = tok()
        if k == "}":
            break
        header[k] = tok()
        assert tok() == ";"
    binary = header["format"] == "binary"
    ncmpt = 3 if header["class"] == "volVectorField" else 1

    def value():
        if ncmpt == 1:
            return float(tok())
        assert tok() == "("
        v = [float(tok()) for _ in range(3)]
        assert tok() == ")"
        return v

    def field_entry():
        kind = tok()
        if kind == "uniform":
            v = value(); assert tok() == ";"
            return ("uniform", v)
        assert kind == "nonuniform"
        t = tok()
        if t.startswith("List<"):
            assert t == ("List<vector>" if ncmpt == 3 else "List<scalar>")
            t = tok()
        cnt = int(t)
        if binary and cnt:
            skip(); assert raw[pos[0]:pos[0] + 1] == b"("
            a = np.frombuffer(raw, dtype=np.float64, count=cnt * ncmpt, offset=pos[0] + 1).copy(); pos[0] += 1 + 8 * cnt * ncmpt
            assert tok() == ")"
        else:
            assert tok() == "("
            a = np.array([value() for _ in range(cnt)], dtype=np.float64)
            assert tok() == ")"
        assert tok() == ";"
        return a.reshape(cnt, 3) if ncmpt == 3 else a.reshape(cnt)

    assert tok() == "dimensions"
    dims = []
    while True:
        t = tok()
        if t == ";":
            break
        dims.append(t)
    assert tok() == "internalField"
    internal = field_entry()
    assert tok() == "boundaryField" and tok() == "{"
    bf = []
    while True:
        name = tok()
        if name == "}":
            break
        assert tok() == "{"
        entries = {}
        while True:
            k = tok()
            if k == "}":
                break
            if k == "value":
                entries[k] = field_entry()
            else:
                entries[k] = tok(); assert tok() == ";"
        bf.append((name, entries))
    return dict(header=header, dimensions=" ".join(dims), internalField=internal, boundaryField=bf, raw=raw)


@pytest.mark.parametrize("fmt, precision", [("ascii", 17), ("binary", 6), ("ascii", 6)])
def test_field_output_round_trip(pkg, tmp_path, fmt, precision):
    """field I/O without a device (polyMeshFoam -roundTrip): <case>/0/S is read, written as <case>/5/S by writeVolScalarField, read again;
    plus the cell centres as a volVectorField.  Binary and 17-digit ascii are exact, 6 digits (IOstream::defaultPrecision) to 6 digits;
    the files are what GeometricField writes -- header, dimensions, internalField via Field::writeEntry, one boundaryField block per patch,
    lists of 11 entries and more one value per line (UListIO.C:110-125), uniform patch values as `uniform v` (Field.C:672-675)."""
    dims = (5, 4, 3)
    pts, faces, owner, neighbour, patches = make_box_mesh(dims)
    n = int(owner.max()) + 1
    G = geometry(pts, faces, owner, neighbour)
    S = np.sin(4 * G["C"][:, 0]) * np.cos(3 * G["C"][:, 1]) + G["C"][:, 2] * 1e3
    case_dir = str(tmp_path / "case")
    write_case(case_dir, pts, faces, owner, neighbour, patches, S, False)
    out = subprocess.run([os.path.join(PKG, "polyMeshFoam"), case_dir, "-roundTrip", "S", "5", fmt, str(precision)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"roundTrip S nCells (\d+) maxAbsDiff (\S+) vector maxAbsDiff (\S+)", out.stdout)
    assert m and int(m.group(1)) == n
    exact = fmt == "binary" or precision >= 17
    tol = 0.0 if exact else 5e-6 * np.abs(S).max()
    assert float(m.group(2)) <= tol and float(m.group(3)) <= (0.0 if exact else 5e-6 * np.abs(G["C"]).max())
    f = read_vol_field(os.path.join(case_dir, "5", "S"))
    assert f["header"] == {**f["header"], "version": "2.0", "format": fmt, "class": "volScalarField", "location": "5", "object": "S"}
    assert ("arch" in f["header"]) == (fmt == "binary")
    assert f["dimensions"] == "[0 0 -1 0 0 0 0]"
    assert (np.array_equal(f["internalField"], S) if exact else np.max(np.abs(f["internalField"] - S)) <= tol)
    for (name, entries), (pname, ptype, cnt, start) in zip(f["boundaryField"], patches):
        assert name == pname
        if ptype == "patch":
            assert entries["type"] == "fixedValue" and entries["value"][0] == "uniform" and entries["value"][1] > 0
        else:
            assert entries == {"type": "zeroGradient"}
    if fmt == "ascii":   # one value per line from 11 entries on, as UList's operator<< writes them
        body = f["raw"].decode()
        assert f"internalField   nonuniform List<scalar> \n{n}\n(\n" in body and "\n)\n;\n" in body
    c = read_vol_field(os.path.join(case_dir, "5", "C"))
    assert c["header"]["class"] == "volVectorField" and c["internalField"].shape == (n, 3)
    assert (np.array_equal(c["internalField"], G["C"]) if exact else np.max(np.abs(c["internalField"] - G["C"])) <= 5e-6 * np.abs(G["C"]).max()) or \
        np.max(np.abs(c["internalField"] - G["C"])) < 1e-12      # (the C++ geometry and numpy's agree to rounding, not to the bit)


def test_reader_rejects_broken_meshes(pkg, tmp_path):
    # error behaviour without a device: the reader validates before anything touches the engine
    dims = (3, 2, 2)
    pts, faces, owner, neighbour, patches = make_box_mesh(dims)
    S = np.zeros(int(owner.max()) + 1)
    bad = patches[:-1] + [(patches[-1][0], patches[-1][1], patches[-1][2] - 1, patches[-1][3])]
    case_dir = str(tmp_path / "bad")
    write_case(case_dir, pts, faces, owner, neighbour, bad, S, False)
    out = subprocess.run([os.path.join(PKG, "polyMeshFoam"), case_dir], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "do not cover the boundary faces" in out.stderr
    out = subprocess.run([os.path.join(PKG, "polyMeshFoam"), str(tmp_path / "missing")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "cannot open file" in out.stderr
    # a cell that owns no face (its faces are all owned by lower cells) still counts: nCells = max over owner AND neighbour
    # (polyMeshInitMesh.C:59-86); negative labels are refused before any geometry is computed
    neg = owner.copy(); neg[0] = -1
    case_dir = str(tmp_path / "neg")
    write_case(case_dir, pts, faces, neg, neighbour, patches, S, False)
    out = subprocess.run([os.path.join(PKG, "polyMeshFoam"), case_dir], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "negative cell label" in out.stderr
    # a binary case written by a 64-bit-label build is a different memory image: refused, not mis-parsed
    case_dir = str(tmp_path / "arch64")
    write_case(case_dir, pts, faces, owner, neighbour, patches, S, True)
    f = os.path.join(case_dir, "constant", "polyMesh", "owner")
    raw = open(f, "rb").read().replace(b"    format      binary;", b"    format      binary;\n    arch        \"LSB;label=64;scalar=64\";", 1)
    open(f, "wb").write(raw)
    out = subprocess.run([os.path.join(PKG, "polyMeshFoam"), case_dir], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "label=64" in out.stderr and "this build reads label=32" in out.stderr
    # the same file with the widths of this build is accepted by the reader (it then needs a device to go on)
    open(f, "wb").write(raw.replace(b"label=64", b"label=32"))
    out = subprocess.run([os.path.join(PKG, "polyMeshFoam"), case_dir], capture_output=True, text=True, timeout=60)
    assert "was written with arch" not in out.stderr


# ---- decomposed cases (processorN directories) -----------------------------------------------------------------------
def decompose_box_mesh(dims, nproc, pts, faces, owner, neighbour, patches):
    """what decomposePar (simple, n = (nproc 1 1)) writes: per processor the local points / faces / owner / neighbour, the
    physical patches (kept, possibly empty) and one processor patch per neighbouring processor, cut faces in global order,
    flipped on the side that holds the global neighbour cell"""
    nx = dims[0]
    n = int(owner.max()) + 1
    proc_of = (np.arange(n) % nx) * nproc // nx
    nI = len(neighbour)
    out = []
    for p in range(nproc):
        cells = np.nonzero(proc_of == p)[0]
        cmap = -np.ones(n, np.int64); cmap[cells] = np.arange(len(cells))
        lf, lo, ln, lp = [], [], [], []
        for f in range(nI):
            if proc_of[owner[f]] == p and proc_of[neighbour[f]] == p:
                lf.append(faces[f]); lo.append(cmap[owner[f]]); ln.append(cmap[neighbour[f]])
        for name, ptype, cnt, start in patches:
            s0 = len(lf)
            for f in range(start, start + cnt):
                if proc_of[owner[f]] == p:
                    lf.append(faces[f]); lo.append(cmap[owner[f]])
            lp.append((name, ptype, len(lf) - s0, s0))
        for q in range(nproc):
            if q == p:
                continue
            s0 = len(lf)
            for f in range(nI):
                a, b = proc_of[owner[f]], proc_of[neighbour[f]]
                if a == p and b == q:
                    lf.append(faces[f]); lo.append(cmap[owner[f]])
                elif a == q and b == p:
                    lf.append(faces[f][::-1]); lo.append(cmap[neighbour[f]])
            if len(lf) > s0:
                lp.append((f"procBoundary{p}to{q}", "processor", len(lf) - s0, s0, p, q))
        lf = np.array(lf, dtype=np.int64)
        used = np.unique(lf)
        pmap = -np.ones(len(pts), np.int64); pmap[used] = np.arange(len(used))
        out.append(dict(pts=pts[used], faces=pmap[lf].astype(np.int32), owner=np.array(lo, np.int32), neighbour=np.array(ln, np.int32),
                        patches=lp, cells=cells))
    return out


CHECK = re.compile(r"processor(\d+): nPoints (\d+) nCells (\d+) nFaces (\d+) nInternalFaces (\d+) sumV (\S+)")


def test_reader_reads_a_decomposed_case(pkg, tmp_path):
    """CPU: processor0/ and processor1/ of a 2-way decomposition in decomposePar's layout; the reader's statistics per
    processor against numpy (cells, faces, volumes, every patch incl. the processor patch entries and their areas)."""
    dims = (8, 5, 4)
    pts, faces, owner, neighbour, patches = make_box_mesh(dims)
    G = geometry(pts, faces, owner, neighbour)
    parts = decompose_box_mesh(dims, 2, pts, faces, owner, neighbour, patches)
    case_dir = str(tmp_path / "dec")
    for p, P in enumerate(parts):
        write_case(os.path.join(case_dir, f"processor{p}"), P["pts"], P["faces"], P["owner"], P["neighbour"], P["patches"], np.zeros(len(P["cells"])), binary=(p == 1))
    cut_area = None
    for p, P in enumerate(parts):
        out = subprocess.run([os.path.join(PKG, "polyMeshFoamPar"), case_dir, "-check", str(p)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        m = CHECK.search(out.stdout)
        assert m and int(m.group(1)) == p
        assert (int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))) == (len(P["pts"]), len(P["cells"]), len(P["faces"]), len(P["neighbour"]))
        assert abs(float(m.group(6)) - G["V"][P["cells"]].sum()) < 1e-12 * G["V"].sum()
        for pt in P["patches"]:
            mm = re.search(rf"patch {pt[0]} type {pt[1]} nFaces {pt[2]} startFace {pt[3]} myProcNo (-?\d+) neighbProcNo (-?\d+) area (\S+)", out.stdout)
            assert mm, (pt, out.stdout)
            if pt[1] == "processor":
                assert (int(mm.group(1)), int(mm.group(2))) == (pt[4], pt[5])
                cut_area = float(mm.group(3)) if cut_area is None else cut_area
                assert abs(float(mm.group(3)) - cut_area) < 1e-13 * cut_area        # both sides see the same faces
    assert sum(len(P["cells"]) for P in parts) == int(owner.max()) + 1


@pytest.mark.gpu
def test_polyMeshFoamPar_solves_a_processor_coupled_case(pkg, orc, tmp_path):
    """One GPU, one rank: processor0/ holds the whole (undistorted) box whose y-min / y-max faces are two `processor` patches
    towards rank 0 itself, so the neighbours' cell centres and every solver's halo travel through RCCL; the coupled
    deltaCoeffs, the assembled Laplacian and both solves are recomputed with numpy + the oracle (cyclic interfaces)."""
    dims = (10, 8, 6)
    nx, ny, nz = dims
    pts, faces, owner, neighbour, patches = make_box_mesh(dims, seed=None)
    nI = len(neighbour)
    (inl, outl, walls) = patches
    w0 = walls[3]
    ymin = np.arange(w0, w0 + nz * nx); ymax = np.arange(w0 + nz * nx, w0 + 2 * nz * nx); zz = np.arange(w0 + 2 * nz * nx, walls[3] + walls[2])
    order = np.concatenate([np.arange(nI), np.arange(inl[3], inl[3] + inl[2]), np.arange(outl[3], outl[3] + outl[2]), zz, ymin, ymax])
    faces2, owner2 = faces[order], owner[order]
    s = nI
    pl = [("inlet", "patch", inl[2], s)]; s += inl[2]
    pl.append(("outlet", "patch", outl[2], s)); s += outl[2]
    pl.append(("walls", "wall", len(zz), s)); s += len(zz)
    sA = s; pl.append(("procBoundary0to0a", "processor", len(ymin), s, 0, 0)); s += len(ymin)
    sB = s; pl.append(("procBoundary0to0b", "processor", len(ymax), s, 0, 0)); s += len(ymax)
    n = int(owner.max()) + 1
    G = geometry(pts, faces2, owner2, neighbour)
    S = np.sin(4 * G["C"][:, 0]) * np.cos(3 * G["C"][:, 1]) + G["C"][:, 2]
    case_dir = str(tmp_path / "self")
    write_case(os.path.join(case_dir, "processor0"), pts, faces2, owner2, neighbour, pl, S, False)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MI_COMM_ID_FILE=str(tmp_path / "ids"))
    out = subprocess.run([os.path.join(PKG, "polyMeshFoamPar"), case_dir], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr + out.stdout
    # coupled geometry: the cell across face i of patch A is the owner of face i of patch B and vice versa
    fcA, fcB = owner2[sA:sA + len(ymin)], owner2[sB:sB + len(ymax)]
    def coupled(start, fc, fn):
        nhat = G["Sf"][start:start + len(fc)] / G["magSf"][start:start + len(fc), None]
        d = G["C"][fn] - G["C"][fc]
        dc = 1.0 / np.maximum((nhat * d).sum(axis=1), 0.05 * np.sqrt((d * d).sum(axis=1)))
        son = (nhat * (G["Cf"][start:start + len(fc)] - G["C"][fc])).sum(axis=1); sne = (nhat * (G["C"][fn] - G["Cf"][start:start + len(fc)])).sum(axis=1)
        return dc, sne / (son + sne)
    dcA, wA = coupled(sA, fcA, fcB); dcB, wB = coupled(sB, fcB, fcA)
    m = re.search(r"coupled geometry: sumDeltaCoeffs (\S+) sumWeights (\S+)", out.stdout)
    assert m, out.stdout
    assert abs(float(m.group(1)) - (dcA.sum() + dcB.sum())) < 1e-12 * (dcA.sum() + dcB.sum())
    assert abs(float(m.group(2)) - (wA.sum() + wB.sum())) < 1e-12 * np.abs(np.concatenate([wA, wB])).sum()
    syn = pkg.synthetic
    lo, up = owner2[:nI], neighbour
    upper, diag = orc.fvm_laplacian(n, lo, up, G["delta"], G["magSf"][:nI])
    for name, ptype, cnt, start in pl[:2]:
        ic = -(G["magSf"][start:start + cnt] * G["delta_b"][start - nI:start - nI + cnt])
        diag = orc.patch_add(owner2[start:start + cnt], ic, diag, 0)
    bouA = -(G["magSf"][sA:sA + len(fcA)] * dcA); bouB = -(G["magSf"][sB:sB + len(fcB)] * dcB)
    diag = orc.patch_add(fcA, bouA, diag, 0); diag = orc.patch_add(fcB, bouB, diag, 0)
    src = S * G["V"]
    case = syn.LduCase(n, lo, up, diag, upper, None, src)
    case.interfaces = [syn.Interface(0, 1, fcA, bouA, bouA), syn.Interface(0, 0, fcB, bouB, bouB)]
    Sys = orc.System([case])
    z = np.zeros(n)
    _, p1 = Sys.pcg(z, src, "AINV", tolerance=1e-9)
    w = np.sqrt(((G["Sf"][:nI] / np.sqrt(G["magSf"][:nI])[:, None] * np.array([1.0, 1.01, 1.02])) ** 2).sum(axis=1))
    x2, p2 = orc.GamgSysHierarchy(Sys, [w], 10).solve(z, src, tolerance=1e-9)
    got = [(mm.group(1), float(mm.group(3)), float(mm.group(4)), int(mm.group(5))) for mm in map(LINE.match, out.stdout.splitlines()) if mm]
    assert [g[0] for g in got] == ["AINVPCG", "GAMG"], out.stdout
    for g, p in zip(got, (p1, p2)):
        assert g[3] == p["nIterations"], (g, p["nIterations"])
        assert abs(g[1] - p["initialResidual"]) < 1e-10 and abs(g[2] - p["finalResidual"]) < 1e-9 * max(p["initialResidual"], 1e-30) + 1e-10
    mm = re.search(r"p sum \(global\) max \(rank 0\): (\S+) (\S+)", out.stdout)
    assert abs(float(mm.group(1)) - x2.sum()) < 1e-6 * np.abs(x2).sum() and abs(float(mm.group(2)) - np.abs(x2).max()) < 1e-6 * np.abs(x2).max()
    assert out.stdout.strip().endswith("End")

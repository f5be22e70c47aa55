// polyMesh.C -- see polyMesh.H
#include "polyMesh.H"

#include <cerrno>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <sys/stat.h>

namespace Foam
{
namespace
{
// a whole OpenFOAM file in memory with a cursor: comments skipped, tokens = words, numbers, punctuation ( ) { } ;
struct IFstream
{
    std::string buf, name;
    std::size_t pos = 0;
    bool binary = false;
    std::string arch;   // FoamFile "arch" entry, if present
    explicit IFstream(const std::string& file) : name(file)
    {
        std::ifstream f(file, std::ios::binary);
        if (!f) FatalErrorIn("IFstream::IFstream", "cannot open file " + file);
        buf.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    }
    void skipSpace()
    {
        for (;;) {
            while (pos < buf.size() && std::isspace((unsigned char)buf[pos])) ++pos;
            if (pos + 1 < buf.size() && buf[pos] == '/' && buf[pos + 1] == '/') { while (pos < buf.size() && buf[pos] != '\n') ++pos; continue; }
            if (pos + 1 < buf.size() && buf[pos] == '/' && buf[pos + 1] == '*') {
                const std::size_t e = buf.find("*/", pos + 2);
                pos = e == std::string::npos ? buf.size() : e + 2; continue;
            }
            return;
        }
    }
    bool eof() { skipSpace(); return pos >= buf.size(); }
    char peek() { skipSpace(); return pos < buf.size() ? buf[pos] : '\0'; }
    std::string token()
    {
        skipSpace();
        if (pos >= buf.size()) FatalErrorIn("IFstream::token", "unexpected end of file " + name);
        const char c = buf[pos];
        if (c == '(' || c == ')' || c == '{' || c == '}' || c == ';') { ++pos; return std::string(1, c); }
        if (c == '"') { const std::size_t e = buf.find('"', pos + 1); std::string s = buf.substr(pos + 1, e - pos - 1); pos = e + 1; return s; }
        const std::size_t b = pos;
        while (pos < buf.size() && !std::isspace((unsigned char)buf[pos]) && !strchr("(){};", buf[pos])) ++pos;
        return buf.substr(b, pos - b);
    }
    void expect(const char* t) { const std::string g = token(); if (g != t) FatalErrorIn("IFstream::expect", "expected '" + std::string(t) + "' but found '" + g + "' in " + name); }
    label readLabel() { return (label)std::strtol(token().c_str(), nullptr, 10); }
    scalar readScalar() { return std::strtod(token().c_str(), nullptr); }
    // FoamFile { ... }: picks up "format"
    void header()
    {
        if (token() != "FoamFile") FatalErrorIn("IFstream::header", name + " does not start with a FoamFile header");
        expect("{");
        for (;;) {
            const std::string k = token();
            if (k == "}") break;
            std::string v = token();
            if (k == "format") binary = (v == "binary");
            else if (k == "arch") arch = v;   // e.g. "LSB;label=32;scalar=64" (written by OpenFOAM >= v1612 / 4.x)
            while (v != ";") v = token();
        }
    }
    // a binary list is a memory image in the writer's label / scalar widths: refuse widths other than this build's
    // (a WM_LABEL_SIZE=64 or single-precision case would otherwise be mis-parsed silently)
    void checkArch() const
    {
        if (arch.empty()) return;          // older writers do not state it: the build's widths are assumed, as OpenFOAM does
        const auto width = [&](const char* key) -> long {
            const std::size_t at = arch.find(key);
            return at == std::string::npos ? -1 : std::strtol(arch.c_str() + at + std::strlen(key), nullptr, 10);
        };
        const long lw = width("label="), sw = width("scalar=");
        if ((lw > 0 && lw != 8 * (long)sizeof(label)) || (sw > 0 && sw != 8 * (long)sizeof(scalar)))
            FatalErrorIn("IFstream::raw", "binary file " + name + " was written with arch \"" + arch + "\"; this build reads label=" +
                         std::to_string(8 * sizeof(label)) + " scalar=" + std::to_string(8 * sizeof(scalar)) + " only");
        if (arch.find("MSB") != std::string::npos) FatalErrorIn("IFstream::raw", "binary file " + name + " is big-endian (arch \"" + arch + "\")");
    }
    // raw bytes of a binary list: the cursor stands right after "N("
    void raw(void* dst, std::size_t bytes)
    {
        checkArch();
        if (pos + bytes > buf.size()) FatalErrorIn("IFstream::raw", "binary list truncated in " + name);
        std::memcpy(dst, buf.data() + pos, bytes); pos += bytes;
    }
};

template <class T, class ReadOne>
std::vector<T> readList(IFstream& is, ReadOne one)
{
    const label n = is.readLabel();
    std::vector<T> out((std::size_t)n);
    if (is.binary && n > 0) {                // "N(<bytes>)": no white space between '(' and the data
        is.skipSpace();
        if (is.buf[is.pos] != '(') FatalErrorIn("readList", "expected '(' in " + is.name);
        ++is.pos;
        is.raw(out.data(), sizeof(T) * (std::size_t)n);
        is.expect(")");
    } else {
        is.expect("(");
        for (label i = 0; i < n; ++i) out[(std::size_t)i] = one(is);
        is.expect(")");
    }
    return out;
}
labelList readLabels(IFstream& is) { return readList<label>(is, [](IFstream& s) { return s.readLabel(); }); }
vector readVec(IFstream& s) { s.expect("("); vector v{s.readScalar(), s.readScalar(), s.readScalar()}; s.expect(")"); return v; }

vector operator-(const vector& a, const vector& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
vector operator+(const vector& a, const vector& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
vector operator*(scalar s, const vector& a) { return {s * a[0], s * a[1], s * a[2]}; }
scalar dot(const vector& a, const vector& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
vector cross(const vector& a, const vector& b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }
scalar mag(const vector& a) { return std::sqrt(dot(a, a)); }
} // namespace

labelList readLabelList(const std::string& file) { IFstream is(file); is.header(); return readLabels(is); }
vectorField readVectorField(const std::string& file) { IFstream is(file); is.header(); return readList<vector>(is, readVec); }

polyMesh::polyMesh(const std::string& caseDir)
{
    const std::string dir = caseDir + "/constant/polyMesh/";
    points = readVectorField(dir + "points");
    owner = readLabelList(dir + "owner");
    neighbour = readLabelList(dir + "neighbour");
    {
        IFstream is(dir + "faces");
        is.header();
        if (is.binary) {                       // faceCompactList: offsets then labels
            const labelList start = readLabels(is), lab = readLabels(is);
            faces.resize(start.size() - 1);
            for (std::size_t f = 0; f + 1 < start.size(); ++f) faces[f].assign(lab.begin() + start[f], lab.begin() + start[f + 1]);
        } else {
            const label n = is.readLabel();
            is.expect("(");
            faces.resize((std::size_t)n);
            for (label f = 0; f < n; ++f) {
                const label np = is.readLabel();
                is.expect("(");
                faces[(std::size_t)f].resize((std::size_t)np);
                for (label k = 0; k < np; ++k) faces[(std::size_t)f][(std::size_t)k] = is.readLabel();
                is.expect(")");
            }
            is.expect(")");
        }
    }
    {
        IFstream is(dir + "boundary");
        is.header();
        const label n = is.readLabel();
        is.expect("(");
        for (label p = 0; p < n; ++p) {
            polyPatch P;
            P.name = is.token();
            is.expect("{");
            for (;;) {
                const std::string k = is.token();
                if (k == "}") break;
                std::vector<std::string> v;
                for (std::string t = is.token(); t != ";"; t = is.token()) v.push_back(t);
                if (v.empty()) continue;
                if (k == "type") P.type = v[0];
                else if (k == "nFaces") P.nFaces = (label)std::atol(v[0].c_str());
                else if (k == "startFace") P.startFace = (label)std::atol(v[0].c_str());
                else if (k == "myProcNo") P.myProcNo = std::atoi(v[0].c_str());
                else if (k == "neighbProcNo") P.neighbProcNo = std::atoi(v[0].c_str());
            }
            boundary.push_back(P);
        }
        is.expect(")");
    }
    if (owner.size() != faces.size()) FatalErrorIn("polyMesh::polyMesh", "owner and faces differ in size");
    if (neighbour.size() > owner.size()) FatalErrorIn("polyMesh::polyMesh", "more neighbours than faces");
    // polyMeshInitMesh.C:59-86: nCells = max over owner AND neighbour (a cell may own no face at all)
    nCells = 0;
    for (label c : owner) { if (c < 0) FatalErrorIn("polyMesh::polyMesh", "negative cell label in owner"); nCells = std::max(nCells, c + 1); }
    for (label c : neighbour) { if (c < 0) FatalErrorIn("polyMesh::polyMesh", "negative cell label in neighbour"); nCells = std::max(nCells, c + 1); }
    for (std::size_t f = 0; f < neighbour.size(); ++f)
        if (!(owner[f] < neighbour[f])) FatalErrorIn("polyMesh::polyMesh", "internal faces are not in upper-triangular order");
    label next = nInternalFaces();
    for (const polyPatch& P : boundary) {
        if (P.startFace != next) FatalErrorIn("polyMesh::polyMesh", "patch " + P.name + " does not start where the previous one ends");
        next += P.nFaces;
    }
    if (next != nFaces()) FatalErrorIn("polyMesh::polyMesh", "boundary patches do not cover the boundary faces");
    calcGeometry();
}

labelList polyMesh::patchFaceCells(label p) const
{
    const polyPatch& P = boundary[(std::size_t)p];
    return labelList(owner.begin() + P.startFace, owner.begin() + P.startFace + P.nFaces);
}

void polyMesh::calcGeometry()
{
    const std::size_t nF = faces.size(), nI = neighbour.size();
    Cf.resize(nF); Sf.resize(nF); magSf.resize(nF);
    for (std::size_t fi = 0; fi < nF; ++fi) {               // primitiveMeshFaceCentresAndAreas.C:60-130
        const labelList& f = faces[fi];
        const std::size_t nP = f.size();
        if (nP == 3) {
            Cf[fi] = (1.0 / 3.0) * (points[f[0]] + points[f[1]] + points[f[2]]);
            Sf[fi] = 0.5 * cross(points[f[1]] - points[f[0]], points[f[2]] - points[f[0]]);
        } else {
            vector sumN{0, 0, 0}, sumAc{0, 0, 0};
            scalar sumA = 0;
            vector fc = points[f[0]];
            for (std::size_t pi = 1; pi < nP; ++pi) fc = fc + points[f[pi]];
            fc = (1.0 / (scalar)nP) * fc;
            for (std::size_t pi = 0; pi < nP; ++pi) {
                const vector& p0 = points[f[pi]];
                const vector& p1 = points[f[(pi + 1) % nP]];
                const vector c = p0 + p1 + fc;
                const vector n = cross(p1 - p0, fc - p0);
                const scalar a = mag(n);
                sumN = sumN + n; sumA += a; sumAc = sumAc + a * c;
            }
            if (sumA < 1e-150) { Cf[fi] = fc; Sf[fi] = {0, 0, 0}; }
            else { Cf[fi] = ((1.0 / 3.0) / sumA) * sumAc; Sf[fi] = 0.5 * sumN; }
        }
        magSf[fi] = mag(Sf[fi]);
    }
    C.assign((std::size_t)nCells, vector{0, 0, 0}); V.assign((std::size_t)nCells, 0.0);   // primitiveMeshCellCentresAndVols.C:60-170
    vectorField cEst((std::size_t)nCells, vector{0, 0, 0});
    std::vector<label> nCellFaces((std::size_t)nCells, 0);
    for (std::size_t f = 0; f < nF; ++f) { cEst[owner[f]] = cEst[owner[f]] + Cf[f]; ++nCellFaces[owner[f]]; }
    for (std::size_t f = 0; f < nI; ++f) { cEst[neighbour[f]] = cEst[neighbour[f]] + Cf[f]; ++nCellFaces[neighbour[f]]; }
    for (label c = 0; c < nCells; ++c) cEst[c] = (1.0 / (scalar)nCellFaces[c]) * cEst[c];
    for (std::size_t f = 0; f < nF; ++f) {
        const label c = owner[f];
        const scalar pyr3 = dot(Sf[f], Cf[f] - cEst[c]);
        C[c] = C[c] + pyr3 * (0.75 * Cf[f] + 0.25 * cEst[c]); V[c] += pyr3;
    }
    for (std::size_t f = 0; f < nI; ++f) {
        const label c = neighbour[f];
        const scalar pyr3 = dot(Sf[f], cEst[c] - Cf[f]);
        C[c] = C[c] + pyr3 * (0.75 * Cf[f] + 0.25 * cEst[c]); V[c] += pyr3;
    }
    for (label c = 0; c < nCells; ++c) {
        if (std::fabs(V[c]) > 1e-300) C[c] = (1.0 / V[c]) * C[c]; else C[c] = cEst[c];
        V[c] *= 1.0 / 3.0;
    }
    weights.resize(nI); nonOrthDeltaCoeffs.resize(nI);               // surfaceInterpolation.C:151-260,368-460
    for (std::size_t f = 0; f < nI; ++f) {
        const scalar sOwn = std::fabs(dot(Sf[f], Cf[f] - C[owner[f]])), sNei = std::fabs(dot(Sf[f], C[neighbour[f]] - Cf[f]));
        weights[f] = sNei / (sOwn + sNei);
        const vector d = C[neighbour[f]] - C[owner[f]];
        const vector n = (1.0 / magSf[f]) * Sf[f];
        nonOrthDeltaCoeffs[f] = 1.0 / std::max(dot(n, d), 0.05 * mag(d));
    }
    nonOrthCorrectionVectors.resize(nI);                             // surfaceInterpolation.C:498-580
    for (std::size_t f = 0; f < nI; ++f) {
        const vector unitArea = (1.0 / magSf[f]) * Sf[f];
        const vector delta = C[neighbour[f]] - C[owner[f]];
        nonOrthCorrectionVectors[f] = unitArea - nonOrthDeltaCoeffs[f] * delta;
    }
    patchDeltaCoeffs.clear(); patchMagSf.clear();
    for (const polyPatch& P : boundary) {
        scalarField dc((std::size_t)P.nFaces), ms((std::size_t)P.nFaces);
        for (label i = 0; i < P.nFaces; ++i) {
            const std::size_t f = (std::size_t)(P.startFace + i);
            const vector d = Cf[f] - C[owner[f]];                    // fvPatch::delta()
            const vector n = (1.0 / magSf[f]) * Sf[f];
            dc[(std::size_t)i] = 1.0 / std::max(dot(n, d), 0.05 * mag(d));
            ms[(std::size_t)i] = magSf[f];
        }
        patchDeltaCoeffs.push_back(dc); patchMagSf.push_back(ms);
    }
}

void polyMesh::coupledPatchGeometry(label patchi, const vectorField& Cn, scalarField& dc, scalarField& w) const
{
    const polyPatch& P = boundary[(std::size_t)patchi];
    if ((label)Cn.size() != P.nFaces) FatalErrorIn("polyMesh::coupledPatchGeometry", "neighbour cell centres do not match patch " + P.name);
    dc.resize((std::size_t)P.nFaces); w.resize((std::size_t)P.nFaces);
    for (label i = 0; i < P.nFaces; ++i) {
        const std::size_t f = (std::size_t)(P.startFace + i);
        const vector n = (1.0 / magSf[f]) * Sf[f];
        const vector d = Cn[(std::size_t)i] - C[owner[f]];
        dc[(std::size_t)i] = 1.0 / std::max(dot(n, d), 0.05 * mag(d));
        const scalar sOwn = dot(n, Cf[f] - C[owner[f]]), sNei = dot(n, Cn[(std::size_t)i] - Cf[f]);
        w[(std::size_t)i] = sNei / (sOwn + sNei);
    }
}

scalarField polyMesh::faceAreaPairWeights() const
{
    // |Sf/sqrt|Sf| o (1, 1.01, 1.02)|  (faceAreaPairGAMGAgglomeration.C:54-81)
    scalarField w(neighbour.size());
    for (std::size_t f = 0; f < neighbour.size(); ++f) {
        const scalar r = 1.0 / std::sqrt(magSf[f]);
        const vector s{Sf[f][0] * r * 1.0, Sf[f][1] * r * 1.01, Sf[f][2] * r * 1.02};
        w[f] = mag(s);
    }
    return w;
}

scalarField readVolScalarInternalField(const std::string& file, label nCells)
{
    IFstream is(file);
    is.header();
    for (;;) {
        if (is.eof()) FatalErrorIn("readVolScalarInternalField", "no internalField in " + file);
        const std::string k = is.token();
        if (k == "internalField") break;
        if (k == "{") { int depth = 1; while (depth) { const std::string t = is.token(); if (t == "{") ++depth; else if (t == "}") --depth; } continue; }
        if (k == "dimensions") { while (is.token() != ";") {} continue; }
    }
    const std::string kind = is.token();
    if (kind == "uniform") return scalarField((std::size_t)nCells, is.readScalar());
    if (kind != "nonuniform") FatalErrorIn("readVolScalarInternalField", "internalField must be uniform or nonuniform in " + file);
    const std::string cls = is.token();
    if (cls != "List<scalar>") FatalErrorIn("readVolScalarInternalField", "expected List<scalar> but found " + cls + " in " + file);
    scalarField v = readList<scalar>(is, [](IFstream& s) { return s.readScalar(); });
    if ((label)v.size() != nCells) FatalErrorIn("readVolScalarInternalField", "field size does not match the mesh in " + file);
    return v;
}

vectorField readVolVectorInternalField(const std::string& file, label nCells)
{
    IFstream is(file);
    is.header();
    for (;;) {
        if (is.eof()) FatalErrorIn("readVolVectorInternalField", "no internalField in " + file);
        const std::string k = is.token();
        if (k == "internalField") break;
        if (k == "{") { int depth = 1; while (depth) { const std::string t = is.token(); if (t == "{") ++depth; else if (t == "}") --depth; } continue; }
        if (k == "dimensions") { while (is.token() != ";") {} continue; }
    }
    const std::string kind = is.token();
    if (kind == "uniform") return vectorField((std::size_t)nCells, readVec(is));
    if (kind != "nonuniform") FatalErrorIn("readVolVectorInternalField", "internalField must be uniform or nonuniform in " + file);
    const std::string cls = is.token();
    if (cls != "List<vector>") FatalErrorIn("readVolVectorInternalField", "expected List<vector> but found " + cls + " in " + file);
    vectorField v = readList<vector>(is, readVec);
    if ((label)v.size() != nCells) FatalErrorIn("readVolVectorInternalField", "field size does not match the mesh in " + file);
    return v;
}

std::vector<patchFieldIn> readVolFieldBoundary(const std::string& file, int nCmpt)
{
    IFstream is(file);
    is.header();
    for (;;) {
        if (is.eof()) FatalErrorIn("readVolFieldBoundary", "no boundaryField in " + file);
        const std::string k = is.token();
        if (k == "boundaryField") break;
        if (k == "dimensions") { while (is.token() != ";") {} continue; }
        if (k == "internalField") {   // skipped: uniform v; | nonuniform List<T> N(...);
            const std::string kind = is.token();
            if (kind == "uniform") { while (is.token() != ";") {} continue; }
            const std::string cls = is.token();
            if (cls == "List<scalar>") (void)readList<scalar>(is, [](IFstream& s) { return s.readScalar(); });
            else if (cls == "List<vector>") (void)readList<vector>(is, readVec);
            else FatalErrorIn("readVolFieldBoundary", "internalField of class " + cls + " in " + file);
            is.expect(";");
        }
    }
    is.expect("{");
    std::vector<patchFieldIn> out;
    for (;;) {
        const std::string name = is.token();
        if (name == "}") break;
        patchFieldIn P; P.patchName = name;
        is.expect("{");
        for (;;) {
            const std::string k = is.token();
            if (k == "}") break;
            if (k == "type") { P.type = is.token(); is.expect(";"); continue; }
            if (k == "value") {
                P.hasValue = true;
                const std::string kind = is.token();
                if (kind == "uniform") {
                    P.uniform = true;
                    if (nCmpt == 1) P.value.push_back(is.readScalar());
                    else { const vector v = readVec(is); P.value.assign(v.begin(), v.end()); }
                } else if (kind == "nonuniform") {
                    std::string cls = is.token();
                    if (cls == "0") { is.expect("("); is.expect(")"); }          // an empty patch: "nonuniform 0()"
                    else if (nCmpt == 1) { if (cls != "List<scalar>") FatalErrorIn("readVolFieldBoundary", "patch " + name + ": expected List<scalar> in " + file); P.value = readList<scalar>(is, [](IFstream& s) { return s.readScalar(); }); }
                    else {
                        if (cls != "List<vector>") FatalErrorIn("readVolFieldBoundary", "patch " + name + ": expected List<vector> in " + file);
                        const vectorField v = readList<vector>(is, readVec);
                        for (const vector& q : v) P.value.insert(P.value.end(), q.begin(), q.end());
                    }
                } else FatalErrorIn("readVolFieldBoundary", "patch " + name + ": value must be uniform or nonuniform in " + file);
                is.expect(";");
                continue;
            }
            while (is.token() != ";") {}      // any other entry of the patch field (inGroups, gradient, ...): skipped
        }
        out.push_back(P);
    }
    return out;
}

namespace
{
// one value of a list: a scalar, or a vector as "(x y z)"
void putValue(std::ostream& os, const scalar* v, int nCmpt)
{
    if (nCmpt == 1) { os << v[0]; return; }
    os << '(';
    for (int d = 0; d < nCmpt; ++d) { if (d) os << ' '; os << v[d]; }
    os << ')';
}
// Field<Type>::writeEntry (Field.C:652-684) over UList's operator<< (UListIO.C:63-135); n values of nCmpt components each
void writeFieldEntry(std::ostream& os, const std::string& indent, const char* keyword, const scalar* v, std::size_t n, int nCmpt, bool binary)
{
    std::string kw = keyword;
    os << indent << kw << std::string(kw.size() < 15 ? 16 - kw.size() : 1, ' ');   // Ostream::writeKeyword: entryIndentation_ = 16
    bool uniform = n > 0;
    for (std::size_t i = 1; uniform && i < n; ++i)
        for (int d = 0; d < nCmpt; ++d) if (v[i * nCmpt + d] != v[d]) { uniform = false; break; }
    if (uniform) { os << "uniform "; putValue(os, v, nCmpt); os << ";\n"; return; }
    os << "nonuniform " << (n ? (nCmpt == 1 ? "List<scalar> " : "List<vector> ") : "");
    if (binary) {
        os << '\n' << n << '\n';
        if (n) { os << '('; os.write(reinterpret_cast<const char*>(v), (std::streamsize)(sizeof(scalar) * n * (std::size_t)nCmpt)); os << ')'; }
    } else if (n <= 1 || n < 11) {
        os << n << '(';
        for (std::size_t i = 0; i < n; ++i) { if (i) os << ' '; putValue(os, v + i * nCmpt, nCmpt); }
        os << ')';
    } else {
        os << '\n' << n << "\n(";
        for (std::size_t i = 0; i < n; ++i) { os << '\n'; putValue(os, v + i * nCmpt, nCmpt); }
        os << "\n)\n";
    }
    os << ";\n";
}
void writeVolField(const std::string& caseDir, const std::string& timeName, const word& object, const char* cls, const std::string& dimensions,
                   const scalar* internal, std::size_t nCells, int nCmpt, const std::vector<patchFieldOut>& boundaryField, bool binary, int precision)
{
    const std::string dir = caseDir + "/" + timeName;
    if (::mkdir(dir.c_str(), 0777) != 0 && errno != EEXIST) FatalErrorIn("writeVolField", "cannot create directory " + dir);
    const std::string file = dir + "/" + object;
    std::ofstream os(file, std::ios::binary);
    if (!os) FatalErrorIn("writeVolField", "cannot open " + file + " for writing");
    os << std::setprecision(precision);
    os << "FoamFile\n{\n    version     2.0;\n    format      " << (binary ? "binary" : "ascii") << ";\n    class       " << cls << ";\n";
    if (binary) os << "    arch        \"LSB;label=" << 8 * sizeof(label) << ";scalar=" << 8 * sizeof(scalar) << "\";\n";
    os << "    location    \"" << timeName << "\";\n    object      " << object << ";\n}\n"
       << "// * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * //\n\n";
    os << "dimensions      " << dimensions << ";\n\n";
    writeFieldEntry(os, "", "internalField", internal, nCells, nCmpt, binary);
    os << "\nboundaryField\n{\n";
    for (const patchFieldOut& P : boundaryField) {
        os << "    " << P.patchName << "\n    {\n        type            " << P.type << ";\n";
        if (P.hasValue) {
            if (P.value.size() % (std::size_t)nCmpt) FatalErrorIn("writeVolField", "patch " + P.patchName + ": value size is not a multiple of the components");
            writeFieldEntry(os, "        ", "value", P.value.data(), P.value.size() / (std::size_t)nCmpt, nCmpt, binary);
        }
        os << "    }\n";
    }
    os << "}\n\n\n// ************************************************************************* //\n";
    if (!os) FatalErrorIn("writeVolField", "write to " + file + " failed");
}
} // namespace

void writeVolScalarField(const std::string& caseDir, const std::string& timeName, const word& object, const std::string& dimensions,
                         const scalarField& internalField, const std::vector<patchFieldOut>& boundaryField, bool binary, int precision)
{
    writeVolField(caseDir, timeName, object, "volScalarField", dimensions, internalField.data(), internalField.size(), 1, boundaryField, binary, precision);
}
void writeVolVectorField(const std::string& caseDir, const std::string& timeName, const word& object, const std::string& dimensions,
                         const vectorField& internalField, const std::vector<patchFieldOut>& boundaryField, bool binary, int precision)
{
    static_assert(sizeof(vector) == 3 * sizeof(scalar), "vectorField is a packed array of 3 scalars");
    writeVolField(caseDir, timeName, object, "volVectorField", dimensions, internalField.empty() ? nullptr : internalField[0].data(), internalField.size(), 3,
                  boundaryField, binary, precision);
}
} // namespace Foam

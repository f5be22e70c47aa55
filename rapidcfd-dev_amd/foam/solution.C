// solution.C -- see solution.H
#include "solution.H"

#include <cstring>
#include <fstream>
#include <regex.h>

namespace Foam
{
namespace
{
struct Lexer
{
    const std::string& s; std::size_t pos = 0; std::string name;
    Lexer(const std::string& text, const std::string& n) : s(text), name(n) {}
    void skip()
    {
        for (;;) {
            while (pos < s.size() && std::isspace((unsigned char)s[pos])) ++pos;
            if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '/') { while (pos < s.size() && s[pos] != '\n') ++pos; continue; }
            if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '*') { const std::size_t e = s.find("*/", pos + 2); pos = e == std::string::npos ? s.size() : e + 2; continue; }
            return;
        }
    }
    bool eof() { skip(); return pos >= s.size(); }
    // one token; quoted = it was a "string" (a pattern when it stands as a key)
    std::string token(bool& quoted)
    {
        skip();
        quoted = false;
        if (pos >= s.size()) FatalErrorIn("dictionary parser", "unexpected end of " + name);
        const char c = s[pos];
        if (c == '{' || c == '}' || c == ';') { ++pos; return std::string(1, c); }
        if (c == '"') {
            const std::size_t e = s.find('"', pos + 1);
            if (e == std::string::npos) FatalErrorIn("dictionary parser", "unterminated string in " + name);
            std::string t = s.substr(pos + 1, e - pos - 1); pos = e + 1; quoted = true; return t;
        }
        const std::size_t b = pos;
        int depth = 0;                      // ( ... ) lists stay one token (e.g. a vector value): balanced parentheses
        while (pos < s.size()) {
            const char d = s[pos];
            if (d == '(') ++depth;
            else if (d == ')') { if (depth == 0) break; --depth; ++pos; if (depth == 0) break; continue; }
            else if (depth == 0 && (std::isspace((unsigned char)d) || d == '{' || d == '}' || d == ';' || d == '"')) break;
            ++pos;
        }
        if (pos == b) FatalErrorIn("dictionary parser", std::string("unexpected '") + s[pos] + "' in " + name);
        return s.substr(b, pos - b);
    }
};

// `$name` / `${name}` looked up in this dictionary and outwards (dictionary::lookupScopedEntryPtr without the scoping dots)
const dictTree::entry* findVar(const dictTree& d, const word& var) { return d.lookupEntryPtr(var, true, true); }

void parseInto(Lexer& lx, dictTree& d, bool top)
{
    for (;;) {
        if (lx.eof()) { if (top) return; FatalErrorIn("dictionary parser", "missing '}' in " + lx.name); }
        bool q = false;
        const std::string k = lx.token(q);
        if (!q && k == "}") { if (top) FatalErrorIn("dictionary parser", "unexpected '}' in " + lx.name); return; }
        if (!q && (k == "{" || k == ";")) FatalErrorIn("dictionary parser", "unexpected '" + k + "' in " + lx.name);
        if (!q && k[0] == '#') FatalErrorIn("dictionary parser", "function entry " + k + " in " + lx.name + ": not supported by this reader (expand it in the case)");
        if (!q && k[0] == '$') {            // `$p;` at dictionary level: the entries of dictionary p are merged in (primitiveEntry::expandVariable)
            word var = k.substr(1);
            if (!var.empty() && var.front() == '{' && var.back() == '}') var = var.substr(1, var.size() - 2);
            const dictTree::entry* e = findVar(d, var);
            if (!e || !e->isDict) FatalErrorIn("dictionary parser", "`" + k + ";` in " + lx.name + ": no dictionary of that name in scope");
            for (const dictTree::entry& src : e->dict->entries) {
                bool replaced = false;
                for (dictTree::entry& mine : d.entries) if (mine.key == src.key) { mine = src; replaced = true; }
                if (!replaced) d.entries.push_back(src);
            }
            bool q2; if (lx.token(q2) != ";") FatalErrorIn("dictionary parser", "expected ';' after " + k + " in " + lx.name);
            continue;
        }
        dictTree::entry e; e.key = k; e.pattern = q;
        bool q2 = false;
        std::string t = lx.token(q2);
        if (!q2 && t == "{") {
            e.isDict = true; e.dict = std::make_shared<dictTree>(); e.dict->parent = &d;
            parseInto(lx, *e.dict, false);
        } else {
            std::string v;
            while (q2 || t != ";") {
                if (!q2 && (t == "{" || t == "}")) FatalErrorIn("dictionary parser", "entry " + k + " in " + lx.name + " is not terminated by ';'");
                if (!q2 && t.size() > 1 && t[0] == '$') {   // a value taken from an earlier entry
                    word var = t.substr(1);
                    if (var.front() == '{' && var.back() == '}') var = var.substr(1, var.size() - 2);
                    const dictTree::entry* src = findVar(d, var);
                    if (!src || src->isDict) FatalErrorIn("dictionary parser", "`" + t + "` in " + lx.name + ": no such entry in scope");
                    t = src->value;
                }
                if (!v.empty()) v += ' ';
                v += q2 ? "\"" + t + "\"" : t;
                t = lx.token(q2);
            }
            e.value = v;
        }
        bool replaced = false;                      // a later entry of the same key replaces the earlier one (dictionary::add with mergeEntry)
        for (dictTree::entry& mine : d.entries) if (mine.key == e.key && mine.pattern == e.pattern) { mine = e; replaced = true; }
        if (!replaced) d.entries.push_back(e);
    }
}

bool wholeMatch(const word& pattern, const word& text)
{
    regex_t re;
    if (regcomp(&re, pattern.c_str(), REG_EXTENDED) != 0) FatalErrorIn("dictTree::lookupEntryPtr", "Failed to compile regular expression '" + pattern + "'");
    regmatch_t m[1];
    const bool ok = regexec(&re, text.c_str(), 1, m, 0) == 0 && m[0].rm_so == 0 && (std::size_t)m[0].rm_eo == text.size();
    regfree(&re);
    return ok;
}
void fixParents(dictTree& d) { for (dictTree::entry& e : d.entries) if (e.isDict) { e.dict->parent = &d; fixParents(*e.dict); } }
} // namespace

const dictTree::entry* dictTree::lookupEntryPtr(const word& keyword, bool recursive, bool patternMatch) const
{
    for (const entry& e : entries) if (!e.pattern && e.key == keyword) return &e;
    if (patternMatch)
        for (std::size_t i = entries.size(); i-- > 0;)     // patternEntries_ is filled at its head: the last pattern added is tried first
            if (entries[i].pattern && wholeMatch(entries[i].key, keyword)) return &entries[i];
    if (recursive && parent) return parent->lookupEntryPtr(keyword, recursive, patternMatch);
    return nullptr;
}
const dictTree& dictTree::subDict(const word& k) const
{
    const entry* e = lookupEntryPtr(k, false, true);
    if (!e) FatalErrorIn("dictionary::subDict(const word& keyword) const", "keyword " + k + " is undefined in dictionary");
    if (!e->isDict) FatalErrorIn("dictionary::subDict(const word& keyword) const", "keyword " + k + " is not a dictionary");
    return *e->dict;
}
word dictTree::lookup(const word& k) const
{
    const entry* e = lookupEntryPtr(k, false, true);
    if (!e) FatalErrorIn("dictionary::lookupEntry(const word&, bool, bool) const", "keyword " + k + " is undefined in dictionary");
    if (e->isDict) FatalErrorIn("dictionary::lookup", "keyword " + k + " is a dictionary");
    return e->value;
}
dictionary dictTree::flat() const
{
    dictionary out;
    for (const entry& e : entries) {
        if (!e.isDict) { out.add(e.key, e.value); continue; }
        for (const entry& s : e.dict->entries) {
            if (s.isDict) continue;
            if (s.key == e.key) out.add(e.key, s.value);
            else out.add(e.key + "." + s.key, s.value);
        }
    }
    return out;
}

std::shared_ptr<dictTree> parseDictionary(const std::string& text, const std::string& nameForErrors)
{
    auto root = std::make_shared<dictTree>();
    Lexer lx(text, nameForErrors);
    parseInto(lx, *root, true);
    fixParents(*root);
    // the FoamFile header is an ordinary sub-dictionary of the file: dropped
    for (std::size_t i = 0; i < root->entries.size(); ++i) if (root->entries[i].key == "FoamFile" && root->entries[i].isDict) { root->entries.erase(root->entries.begin() + (std::ptrdiff_t)i); break; }
    return root;
}
std::shared_ptr<dictTree> readDictionaryFile(const std::string& file)
{
    std::ifstream f(file, std::ios::binary);
    if (!f) FatalErrorIn("readDictionaryFile", "cannot open file " + file);
    const std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return parseDictionary(text, file);
}

solution::solution(const std::string& caseDir, const std::string& dictName) : dict_(readDictionaryFile(caseDir + "/system/" + dictName)) { read(); }
solution::solution(const std::string& text, const std::string& nameForErrors, int) : dict_(parseDictionary(text, nameForErrors)) { read(); }

void solution::read()
{
    sol_ = dict_->found("select") ? &dict_->subDict(dict_->lookup("select")) : dict_.get();
    fieldRelaxDict_ = std::make_shared<dictTree>(); eqnRelaxDict_ = std::make_shared<dictTree>();
    if (sol_->found("relaxationFactors")) {
        const dictTree& relaxDict = sol_->subDict("relaxationFactors");
        if (relaxDict.found("fields") || relaxDict.found("equations")) {
            if (relaxDict.found("fields")) *fieldRelaxDict_ = relaxDict.subDict("fields");
            if (relaxDict.found("equations")) *eqnRelaxDict_ = relaxDict.subDict("equations");
        } else {   // backwards compatibility (solution.C:77-101): names starting with p or rho are field factors, everything is an equation factor
            for (const dictTree::entry& e : relaxDict.entries) {
                if (e.isDict) continue;
                if (e.key.substr(0, 1) == "p" || (e.key.size() >= 3 && e.key.substr(0, 3) == "rho")) fieldRelaxDict_->entries.push_back(e);
            }
            *eqnRelaxDict_ = relaxDict;
        }
        fieldRelaxDict_->parent = nullptr; eqnRelaxDict_->parent = nullptr;
        fieldRelaxDefault_ = fieldRelaxDict_->found("default") ? std::strtod(fieldRelaxDict_->lookup("default").c_str(), nullptr) : 0.0;
        eqnRelaxDefault_ = eqnRelaxDict_->found("default") ? std::strtod(eqnRelaxDict_->lookup("default").c_str(), nullptr) : 0.0;
    }
}
dictionary solution::solverDict(const word& name) const { return sol_->subDict("solvers").subDict(name).flat(); }
bool solution::relaxField(const word& name) const { return fieldRelaxDict_->found(name) || fieldRelaxDict_->found("default"); }
bool solution::relaxEquation(const word& name) const { return eqnRelaxDict_->found(name) || eqnRelaxDict_->found("default"); }
scalar solution::fieldRelaxationFactor(const word& name) const
{
    if (fieldRelaxDict_->found(name)) return std::strtod(fieldRelaxDict_->lookup(name).c_str(), nullptr);
    if (fieldRelaxDefault_ > 1e-15) return fieldRelaxDefault_;
    FatalErrorIn("Foam::solution::fieldRelaxationFactor(const word&)", "Cannot find variable relaxation factor for '" + name + "' or a suitable default value.");
}
scalar solution::equationRelaxationFactor(const word& name) const
{
    if (eqnRelaxDict_->found(name)) return std::strtod(eqnRelaxDict_->lookup(name).c_str(), nullptr);
    if (eqnRelaxDefault_ > 1e-15) return eqnRelaxDefault_;
    FatalErrorIn("Foam::solution::eqnRelaxationFactor(const word&)", "Cannot find equation relaxation factor for '" + name + "' or a suitable default value.");
}

// ---- fvSchemes ---------------------------------------------------------------------------------------------------------------
namespace
{
wordList splitTokens(const word& v)
{
    wordList t; std::size_t b = 0;
    while (b < v.size()) {
        while (b < v.size() && v[b] == ' ') ++b;
        std::size_t e = b;
        while (e < v.size() && v[e] != ' ') ++e;
        if (e > b) t.push_back(v.substr(b, e - b));
        b = e;
    }
    return t;
}
}
fvSchemes::fvSchemes(const std::string& caseDir, const std::string& dictName) : dict_(readDictionaryFile(caseDir + "/system/" + dictName)) { read(); }
fvSchemes::fvSchemes(const std::string& text, const std::string& nameForErrors, int) : dict_(parseDictionary(text, nameForErrors)) { read(); }
void fvSchemes::readKind(kind& k, const char* name, bool setNoneWhenAbsent)
{
    k.d = std::make_shared<dictTree>();
    if (sch_->found(name)) *k.d = sch_->subDict(name);
    else if (setNoneWhenAbsent) { dictTree::entry e; e.key = "default"; e.value = "none"; k.d->entries.push_back(e); }   // (ddtSchemes / d2dt2Schemes: fvSchemes.C:100-103,132-135)
    k.d->parent = nullptr;
    k.def.clear();
    if (k.d->found("default")) { const word v = k.d->lookup("default"); if (splitTokens(v).empty() || splitTokens(v)[0] != "none") k.def = v; }
}
void fvSchemes::read()
{
    sch_ = dict_->found("select") ? &dict_->subDict(dict_->lookup("select")) : dict_.get();   // fvSchemes.C:411-421
    readKind(ddt_, "ddtSchemes", true);
    if (!sch_->found("ddtSchemes") && sch_->found("timeScheme")) {   // backward compatibility, fvSchemes.C:64-99
        word n = sch_->lookup("timeScheme");
        if (n == "EulerImplicit") n = "Euler"; else if (n == "BackwardDifferencing") n = "backward"; else if (n == "SteadyState") n = "steadyState";
        else FatalErrorIn("fvSchemes::read()", "\n    Only EulerImplicit, BackwardDifferencing and SteadyState\n    are supported by the old timeScheme specification.\n    Please use ddtSchemes instead.");
        ddt_.d->entries.clear(); dictTree::entry e; e.key = "default"; e.value = n; ddt_.d->entries.push_back(e); ddt_.def = n;
    }
    steady_ = ddt_.def == "steadyState";
    readKind(d2dt2_, "d2dt2Schemes", true);
    readKind(interpolation_, "interpolationSchemes", false);
    readKind(div_, "divSchemes", false);
    readKind(grad_, "gradSchemes", false);
    readKind(snGrad_, "snGradSchemes", false);
    readKind(laplacian_, "laplacianSchemes", false);
    fluxRequired_ = std::make_shared<dictTree>();
    if (sch_->found("fluxRequired")) {
        *fluxRequired_ = sch_->subDict("fluxRequired"); fluxRequired_->parent = nullptr;
        if (fluxRequired_->found("default")) { const word v = fluxRequired_->lookup("default"); defaultFluxRequired_ = v != "none" && (v == "yes" || v == "true" || v == "on"); }
    }
}
wordList fvSchemes::lookupIn(const kind& k, const word& name)
{
    if (k.d->found(name) || k.def.empty()) return splitTokens(k.d->lookup(name));   // (lookup: the reference's "keyword ... is undefined in dictionary")
    return splitTokens(k.def);
}
} // namespace Foam

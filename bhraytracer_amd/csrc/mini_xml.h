// mini_xml.h — a small DOM reader for the subset of XML the reference's scene files use
// (elements, attributes, comments, <?...?> / <!...> declarations, self-closing tags).
// It replaces the vendored tinyxml2 the reference parses scenes with (xmlload.cpp:65-72);
// only the calls xmlload makes are mirrored: first child / next sibling element, element
// name, string / double / int attribute queries (a failed query leaves the value untouched,
// like XMLElement::QueryDoubleAttribute).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace bhrt {

struct XmlElement {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlElement>> children;

    const char *Attribute(const char *key) const
    {
        for (auto &a : attrs)
            if (a.first == key) return a.second.c_str();
        return nullptr;
    }
    // tinyxml2 XMLUtil::ToDouble is sscanf("%lf"): success iff a number prefix parses
    bool QueryDouble(const char *key, double *out) const
    {
        const char *s = Attribute(key);
        if (!s) return false;
        char *end = nullptr;
        double v = strtod(s, &end);
        if (end == s) return false;
        *out = v;
        return true;
    }
    bool QueryInt(const char *key, int *out) const
    {
        const char *s = Attribute(key);
        if (!s) return false;
        char *end = nullptr;
        long v = strtol(s, &end, 10);
        if (end == s) return false;
        *out = (int)v;
        return true;
    }
    const XmlElement *FirstChild(const char *nm = nullptr) const
    {
        for (auto &c : children)
            if (!nm || c->name == nm) return c.get();
        return nullptr;
    }
};

class XmlDocument {
public:
    std::vector<std::unique_ptr<XmlElement>> roots;
    std::string error;

    bool LoadFile(const char *path)
    {
        FILE *fp = fopen(path, "rb");
        if (!fp) { error = "cannot open file"; return false; }
        std::string text;
        char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
        fclose(fp);
        return Parse(text);
    }

    bool Parse(const std::string &text)
    {
        s_ = text.c_str();
        p_ = 0;
        n_ = text.size();
        roots.clear();
        error.clear();
        while (true) {
            SkipMisc();
            if (p_ >= n_) break;
            if (s_[p_] != '<') { error = "text outside of an element"; return false; }
            std::unique_ptr<XmlElement> e(new XmlElement);
            if (!ParseElement(*e)) return false;
            roots.push_back(std::move(e));
        }
        return true;
    }

    const XmlElement *FirstChild(const char *nm) const
    {
        for (auto &c : roots)
            if (c->name == nm) return c.get();
        return nullptr;
    }

private:
    const char *s_ = nullptr;
    size_t p_ = 0, n_ = 0;

    static bool IsSpace(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
    static bool IsNameChar(char c) { return !(IsSpace(c) || c == '>' || c == '/' || c == '=' || c == '<' || c == '\0'); }
    void SkipSpace() { while (p_ < n_ && IsSpace(s_[p_])) p_++; }
    bool StartsWith(const char *t) const { size_t l = strlen(t); return p_ + l <= n_ && strncmp(s_ + p_, t, l) == 0; }
    // skips whitespace, text, comments, processing instructions and declarations
    void SkipMisc()
    {
        while (p_ < n_) {
            if (StartsWith("<!--")) {
                const char *e = strstr(s_ + p_ + 4, "-->");
                p_ = e ? (size_t)(e - s_) + 3 : n_;
            } else if (StartsWith("<?")) {
                const char *e = strstr(s_ + p_ + 2, "?>");
                p_ = e ? (size_t)(e - s_) + 2 : n_;
            } else if (StartsWith("<!")) {
                const char *e = strchr(s_ + p_ + 2, '>');
                p_ = e ? (size_t)(e - s_) + 1 : n_;
            } else if (s_[p_] == '<') {
                return;
            } else {
                p_++; // character data between elements is ignored (the scene schema has none)
            }
        }
    }
    static std::string Unescape(const std::string &v)
    {
        if (v.find('&') == std::string::npos) return v;
        std::string o;
        for (size_t i = 0; i < v.size(); i++) {
            if (v[i] == '&') {
                if (v.compare(i, 4, "&lt;") == 0) { o += '<'; i += 3; continue; }
                if (v.compare(i, 4, "&gt;") == 0) { o += '>'; i += 3; continue; }
                if (v.compare(i, 5, "&amp;") == 0) { o += '&'; i += 4; continue; }
                if (v.compare(i, 6, "&quot;") == 0) { o += '"'; i += 5; continue; }
                if (v.compare(i, 6, "&apos;") == 0) { o += '\''; i += 5; continue; }
            }
            o += v[i];
        }
        return o;
    }
    bool ParseElement(XmlElement &e)
    {
        p_++; // '<'
        size_t b = p_;
        while (p_ < n_ && IsNameChar(s_[p_])) p_++;
        e.name.assign(s_ + b, p_ - b);
        if (e.name.empty()) { error = "empty element name"; return false; }
        while (true) { // attributes
            SkipSpace();
            if (p_ >= n_) { error = "unexpected end inside tag"; return false; }
            if (s_[p_] == '/') {
                if (p_ + 1 < n_ && s_[p_ + 1] == '>') { p_ += 2; return true; }
                error = "stray '/' in tag"; return false;
            }
            if (s_[p_] == '>') { p_++; break; }
            size_t ab = p_;
            while (p_ < n_ && IsNameChar(s_[p_])) p_++;
            std::string key(s_ + ab, p_ - ab);
            SkipSpace();
            if (p_ >= n_ || s_[p_] != '=') { error = "attribute without value: " + key; return false; }
            p_++;
            SkipSpace();
            if (p_ >= n_ || (s_[p_] != '"' && s_[p_] != '\'')) { error = "unquoted attribute value: " + key; return false; }
            char q = s_[p_++];
            size_t vb = p_;
            while (p_ < n_ && s_[p_] != q) p_++;
            if (p_ >= n_) { error = "unterminated attribute value"; return false; }
            e.attrs.emplace_back(key, Unescape(std::string(s_ + vb, p_ - vb)));
            p_++;
        }
        while (true) { // children
            SkipMisc();
            if (p_ >= n_) { error = "missing </" + e.name + ">"; return false; }
            if (StartsWith("</")) {
                p_ += 2;
                size_t cb = p_;
                while (p_ < n_ && IsNameChar(s_[p_])) p_++;
                std::string close(s_ + cb, p_ - cb);
                SkipSpace();
                if (p_ < n_ && s_[p_] == '>') p_++;
                if (close != e.name) { error = "mismatched </" + close + "> for <" + e.name + ">"; return false; }
                return true;
            }
            std::unique_ptr<XmlElement> c(new XmlElement);
            if (!ParseElement(*c)) return false;
            e.children.push_back(std::move(c));
        }
    }
};

} // namespace bhrt

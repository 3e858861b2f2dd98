// Minimal readers for the two text formats the reference is configured with:
//   - Boost.PropertyTree INFO (task.info / reference.info / gait.info; the reference reads them through
//     ocs2::loadData, e.g. qm_interface/src/QMInterface.cpp:65-73,85,155-156 and qm_interface/src/common/ModelSettings.cpp:18-34)
//   - URDF XML (qm_description/urdf/quadruped_manipulator/robot.urdf, parsed by urdfdom inside
//     ocs2 centroidal_model::createPinocchioInterface, qm_interface/src/QMInterface.cpp:410-411)
// Host-only, dependency-free; errors are reported with std::runtime_error and translated to status codes at the C ABI.
#pragma once
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace qmhost {

inline std::string readFile(const std::string& path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::invalid_argument("file not found: " + path);
  std::ostringstream ss;
  ss << in.rdbuf();
  return ss.str();
}

// ------------------------------------------------------------------------------------------------ INFO
struct InfoNode {
  std::string key, value;
  std::vector<InfoNode> children;
  const InfoNode* child(const std::string& k) const {
    for (const auto& c : children) if (c.key == k) return &c;
    return nullptr;
  }
  // dotted path lookup, e.g. "jointVelocityLimits.lowerBound.arm"
  const InfoNode* find(const std::string& path) const {
    const InfoNode* n = this;
    size_t pos = 0;
    while (n && pos <= path.size()) {
      const size_t dot = path.find('.', pos);
      const std::string k = path.substr(pos, dot == std::string::npos ? std::string::npos : dot - pos);
      n = n->child(k);
      if (dot == std::string::npos) break;
      pos = dot + 1;
    }
    return n;
  }
};

class InfoParser {
 public:
  static InfoNode parseFile(const std::string& path) { return InfoParser(readFile(path)).parse(); }
  explicit InfoParser(std::string text) : s_(std::move(text)) {}
  InfoNode parse() {
    InfoNode root;
    parseBlock(root, /*top=*/true);
    return root;
  }

 private:
  std::string s_;
  size_t p_ = 0;
  // token kinds: word, '{', '}', newline, end
  enum Kind { Word, Open, Close, Newline, End };
  Kind next(std::string& out) {
    while (p_ < s_.size()) {
      const char c = s_[p_];
      if (c == '\n') { ++p_; return Newline; }
      if (std::isspace(static_cast<unsigned char>(c))) { ++p_; continue; }
      if (c == ';' || (c == '/' && p_ + 1 < s_.size() && s_[p_ + 1] == '/')) { while (p_ < s_.size() && s_[p_] != '\n') ++p_; continue; }
      if (c == '{') { ++p_; return Open; }
      if (c == '}') { ++p_; return Close; }
      if (c == '"') {
        const size_t e = s_.find('"', p_ + 1);
        if (e == std::string::npos) throw std::runtime_error("INFO: unterminated string");
        out = s_.substr(p_ + 1, e - p_ - 1);
        p_ = e + 1;
        return Word;
      }
      size_t e = p_;
      while (e < s_.size() && !std::isspace(static_cast<unsigned char>(s_[e])) && s_[e] != '{' && s_[e] != '}' && s_[e] != ';') ++e;
      out = s_.substr(p_, e - p_);
      p_ = e;
      return Word;
    }
    return End;
  }
  void parseBlock(InfoNode& parent, bool top) {
    std::string tok;
    for (;;) {
      Kind k = next(tok);
      if (k == Newline) continue;
      if (k == End) { if (!top) throw std::runtime_error("INFO: missing '}'"); return; }
      if (k == Close) { if (top) throw std::runtime_error("INFO: unmatched '}'"); return; }
      if (k == Open) throw std::runtime_error("INFO: '{' without a key");
      InfoNode node;
      node.key = tok;
      // optional value on the same line
      size_t save = p_;
      k = next(tok);
      if (k == Word) { node.value = tok; save = p_; k = next(tok); }
      while (k == Newline) { save = p_; k = next(tok); }
      if (k == Open) parseBlock(node, false);
      else p_ = save;  // un-read
      parent.children.push_back(std::move(node));
    }
  }
};

inline double infoDouble(const InfoNode& root, const std::string& path) {
  const InfoNode* n = root.find(path);
  if (!n || n->value.empty()) throw std::runtime_error("INFO: missing key '" + path + "'");
  char* end = nullptr;
  const double v = std::strtod(n->value.c_str(), &end);
  if (end == n->value.c_str()) throw std::runtime_error("INFO: key '" + path + "' is not a number: " + n->value);
  return v;
}
inline double infoDoubleOr(const InfoNode& root, const std::string& path, double dflt) {
  const InfoNode* n = root.find(path);
  return (n && !n->value.empty()) ? infoDouble(root, path) : dflt;
}
// ocs2::loadData::loadEigenMatrix semantics: optional "scaling", entries "(i,j) value", missing entries are zero.
inline void infoMatrix(const InfoNode& root, const std::string& path, int rows, int cols, double* out /*row major*/) {
  const InfoNode* n = root.find(path);
  if (!n) throw std::runtime_error("INFO: missing matrix '" + path + "'");
  double scaling = 1.0;
  if (const InfoNode* s = n->child("scaling")) scaling = std::strtod(s->value.c_str(), nullptr);
  for (int i = 0; i < rows * cols; ++i) out[i] = 0.0;
  for (const auto& c : n->children) {
    if (c.key.size() < 5 || c.key.front() != '(') continue;
    int i = -1, j = -1;
    if (std::sscanf(c.key.c_str(), "(%d,%d)", &i, &j) != 2) continue;
    if (i < 0 || j < 0 || i >= rows || j >= cols) continue;
    out[i * cols + j] = scaling * std::strtod(c.value.c_str(), nullptr);
  }
}

// ------------------------------------------------------------------------------------------------ XML
struct XmlNode {
  std::string name;
  std::map<std::string, std::string> attr;
  std::vector<std::unique_ptr<XmlNode>> children;
  const XmlNode* child(const std::string& n) const {
    for (const auto& c : children) if (c->name == n) return c.get();
    return nullptr;
  }
  std::string get(const std::string& k, const std::string& dflt = "") const {
    auto it = attr.find(k);
    return it == attr.end() ? dflt : it->second;
  }
};

class XmlParser {
 public:
  static std::unique_ptr<XmlNode> parseFile(const std::string& path) { return XmlParser(readFile(path)).parse(); }
  explicit XmlParser(std::string text) : s_(std::move(text)) {}
  std::unique_ptr<XmlNode> parse() {
    auto root = std::make_unique<XmlNode>();
    std::vector<XmlNode*> stack{root.get()};
    while (p_ < s_.size()) {
      const size_t lt = s_.find('<', p_);
      if (lt == std::string::npos) break;
      p_ = lt;
      if (s_.compare(p_, 4, "<!--") == 0) { const size_t e = s_.find("-->", p_); if (e == std::string::npos) throw std::runtime_error("XML: unterminated comment"); p_ = e + 3; continue; }
      if (s_.compare(p_, 2, "<?") == 0) { const size_t e = s_.find("?>", p_); if (e == std::string::npos) throw std::runtime_error("XML: unterminated declaration"); p_ = e + 2; continue; }
      if (s_.compare(p_, 2, "<!") == 0) { const size_t e = s_.find('>', p_); p_ = e + 1; continue; }
      if (s_.compare(p_, 2, "</") == 0) {
        const size_t e = s_.find('>', p_);
        if (stack.size() <= 1) throw std::runtime_error("XML: unmatched closing tag");
        stack.pop_back();
        p_ = e + 1;
        continue;
      }
      ++p_;
      auto node = std::make_unique<XmlNode>();
      while (p_ < s_.size() && !std::isspace(static_cast<unsigned char>(s_[p_])) && s_[p_] != '>' && s_[p_] != '/') node->name += s_[p_++];
      bool selfClose = false;
      for (;;) {
        while (p_ < s_.size() && std::isspace(static_cast<unsigned char>(s_[p_]))) ++p_;
        if (p_ >= s_.size()) throw std::runtime_error("XML: unterminated tag");
        if (s_[p_] == '/') { selfClose = true; ++p_; continue; }
        if (s_[p_] == '>') { ++p_; break; }
        std::string key;
        while (p_ < s_.size() && s_[p_] != '=' && !std::isspace(static_cast<unsigned char>(s_[p_]))) key += s_[p_++];
        while (p_ < s_.size() && (std::isspace(static_cast<unsigned char>(s_[p_])) || s_[p_] == '=')) ++p_;
        const char quote = s_[p_];
        if (quote != '"' && quote != '\'') throw std::runtime_error("XML: attribute without quotes in <" + node->name + ">");
        const size_t e = s_.find(quote, p_ + 1);
        if (e == std::string::npos) throw std::runtime_error("XML: unterminated attribute");
        node->attr[key] = s_.substr(p_ + 1, e - p_ - 1);
        p_ = e + 1;
      }
      XmlNode* raw = node.get();
      stack.back()->children.push_back(std::move(node));
      if (!selfClose) stack.push_back(raw);
    }
    if (stack.size() != 1) throw std::runtime_error("XML: unclosed element <" + stack.back()->name + ">");
    return root;
  }

 private:
  std::string s_;
  size_t p_ = 0;
};

inline void parseTriple(const std::string& s, double out[3], const char* what) {
  std::istringstream is(s);
  if (!(is >> out[0] >> out[1] >> out[2])) throw std::runtime_error(std::string("URDF: bad triple for ") + what + ": '" + s + "'");
}

}  // namespace qmhost

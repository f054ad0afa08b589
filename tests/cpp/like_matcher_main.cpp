// CPU-only driver for the host mirror's LikeMatcher (hyrise_amd/host/hyrise_host.hpp): reads lines "condition<TAB>pattern<TAB>text"
// (hex-encoded pattern and text), prints 0 / 1 per line.  tests/test_host_like_matcher.py compares it with hyrise_amd/like.py.
#include <cstdio>
#include <iostream>
#include <string>

#include "../../hyrise_amd/host/hyrise_host.hpp"

static std::string unhex(const std::string& hex) {
  std::string out;
  for (size_t i = 0; i + 1 < hex.size(); i += 2) out.push_back(static_cast<char>(std::stoi(hex.substr(i, 2), nullptr, 16)));
  return out;
}

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    const auto first = line.find('\t'), second = line.find('\t', first + 1);
    const int condition = std::stoi(line.substr(0, first));
    const std::string pattern = unhex(line.substr(first + 1, second - first - 1)), text = unhex(line.substr(second + 1));
    const hyrise_amd::LikeMatcher matcher(pattern, static_cast<hyrise_amd::PredicateCondition>(condition));
    std::puts(matcher(text) ? "1" : "0");
  }
  return 0;
}

// Sanitizer harness (tools/sanitize/fuzz_parser.sh): every file named on the command line through the parser and all its chunk decoders.
// Rejections are expected (the files are mutated on purpose); what must not happen is a sanitizer report.
#include <cstdio>
#include <string>

#include "parser.h"

using namespace glz;

int main(int argc, char** argv) {
  int parsed = 0, rejected = 0;
  for (int i = 1; i < argc; ++i) {
    Error err;
    auto p = Parsed::open(argv[i], err);
    if (!p) { ++rejected; continue; }
    SceneData sd;
    if (!p->to_scene_data(sd, err)) { ++rejected; continue; }
    for (auto& t : sd.textures) { std::string e; (void)t.decode_more_levels(e); }
    ++parsed;
  }
  printf("parsed %d, rejected %d\n", parsed, rejected);
  return 0;
}

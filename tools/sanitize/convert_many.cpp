// Sanitizer harness (tools/sanitize/fuzz_converter.sh): every .obj named on the command line through the converter.
#include <cstdio>
#include <string>

#include "converter.h"

using namespace glz;

int main(int argc, char** argv) {
  int converted = 0, rejected = 0;
  const std::string out = argv[1];
  for (int i = 2; i < argc; ++i) {
    Error err;
    ConvertReport rep;
    if (convert_obj(argv[i], out, (i & 1) != 0, &rep, err)) ++converted; else ++rejected;
  }
  printf("converted %d, rejected %d\n", converted, rejected);
  return 0;
}

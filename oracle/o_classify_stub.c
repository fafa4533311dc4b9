#include "o_classify.h"
int o_classify_main(int argc, char **argv) { (void)argc; (void)argv; fprintf(stderr, "verb not built yet\n"); return 2; }

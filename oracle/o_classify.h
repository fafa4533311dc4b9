/* oracle/ — TEST INFRASTRUCTURE ONLY (see o_common.h). Classify-stage restatement. */
#ifndef PGX_ORACLE_CLASSIFY_H
#define PGX_ORACLE_CLASSIFY_H
#include "o_common.h"
#ifdef __cplusplus
extern "C" {
#endif
int o_classify_main(int argc, char **argv);
#ifdef __cplusplus
}
#endif
#endif

/* panda_chain.c -- CPU ORACLE (test infrastructure). Placeholder, filled in below. */
#include "m3_oracle.h"

// k_sweep4s for windows of 21 band groups (81..84 bands): see cmf_loocv4_nj.inc
#define SW4S_NJ 21
#include "cmf_loocv4_nj.inc"

// k_sweep4s for windows of 24 band groups (93..96 bands): see cmf_loocv4_nj.inc
#define SW4S_NJ 24
#include "cmf_loocv4_nj.inc"

// am_cross_attn_f16.hip — the IEEE-half build of am_cross_attn.hip (see elem16.h): the same source compiled with elem_t = _Float16
#define RL4CO_ELEM_F16 1
#include "am_cross_attn.hip"

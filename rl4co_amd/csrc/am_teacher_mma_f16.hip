// am_teacher_mma_f16.hip — the IEEE-half build of am_teacher_mma.hip (see elem16.h): the same source compiled with elem_t = _Float16
// (fp16 planes, queries, softmax numerators and glimpses; launcher rl4co::launch_*_f16), for the reference's default
// "16-mixed" precision (rl4co/utils/trainer.py:57).
#define RL4CO_ELEM_F16 1
#include "am_teacher_mma.hip"

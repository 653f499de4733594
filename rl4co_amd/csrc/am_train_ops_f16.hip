// am_train_ops_f16.hip — the IEEE-half build of am_train_ops.hip (see elem16.h): the same source compiled with elem_t = _Float16
// and entry points rl4co_*_f16, for the reference's default "16-mixed" precision (rl4co/utils/trainer.py:57).
#define RL4CO_ELEM_F16 1
#include "am_train_ops.hip"

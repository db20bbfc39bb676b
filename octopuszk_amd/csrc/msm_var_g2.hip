// Variable-base MSM: the G2 (Fq2) instantiation of the driver templates and of every kernel they
// launch — a translation unit of its own so that it compiles next to the G1 one.
#include "msm_var_driver.cuh"

OZK_G2_DRIVER_INSTANCES()

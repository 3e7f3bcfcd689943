// SGD kernel instantiations for row-group shape G=16 lanes x KPL=8 dwords per lane (see rfm_sgd.hpp)
#define RFM_G 16
#define RFM_KPL 8
#define RFM_SHAPE_FN sgd_table_g16_k8
#include "rfm_sgd_inst.inc"

#pragma once  // oracle/ref_shim: no registry (layers are constructed directly by oracle/ref_shim/ref_api.cpp)
#define REGISTER_LAYER_CLASS(type)

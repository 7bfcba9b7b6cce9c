#pragma once  // oracle/ref_shim: CPU_ONLY stubs live in common.hpp

#pragma once  // oracle/ref_shim: nothing of SyncedMemory is needed by the files compiled here

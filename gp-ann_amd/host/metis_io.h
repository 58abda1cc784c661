// metis_io.h - forwards to the host mirror (reference header of the same name: src/metis_io.h)
#pragma once
#include "gpann_host.h"

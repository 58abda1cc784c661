// inverted_index.h - forwards to the host mirror (reference header of the same name: src/inverted_index.h)
#pragma once
#include "gpann_host.h"

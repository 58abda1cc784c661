// topn.h - forwards to the host mirror (reference header of the same name: src/topn.h)
#pragma once
#include "gpann_host.h"

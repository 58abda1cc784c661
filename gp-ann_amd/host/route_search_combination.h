// route_search_combination.h - forwards to the host mirror (reference header of the same name: src/route_search_combination.h)
#pragma once
#include "gpann_host.h"

// kmeans_tree_router.h - forwards to the host mirror (reference header of the same name: src/kmeans_tree_router.h)
#pragma once
#include "gpann_host.h"

// ABI plumbing shared by every entry point: version, thread-local error string, cfg validation.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void dgppo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int32_t dgppo_abi_version(void) { return DGPPO_ABI_VERSION; }
extern "C" const char* dgppo_last_error(void) { return g_err; }

int32_t dgppo_validate_cfg(const dgppo_env_cfg* c) {
  DGPPO_REQUIRE(c != nullptr, "cfg is NULL");
  DGPPO_REQUIRE(c->kind >= DGPPO_ENV_LIDAR_SPREAD && c->kind <= DGPPO_ENV_MPE_CONNECT_SPREAD, "unknown env kind %d", c->kind);
  DGPPO_REQUIRE(c->n_agents >= 1 && c->n_agents <= 64, "n_agents must be in [1,64] (got %d)", c->n_agents);
  {
    const bool line = c->kind == DGPPO_ENV_LIDAR_LINE || c->kind == DGPPO_ENV_MPE_LINE;
    const int want_goals = line ? 2 : (c->kind == DGPPO_ENV_MPE_FORMATION ? 1 : c->n_agents);
    DGPPO_REQUIRE(c->n_goals == want_goals, "n_goals must be %d for env kind %d (got %d)", want_goals, c->kind, c->n_goals);
    const int want_rg = c->kind == DGPPO_ENV_LIDAR_LINE ? DGPPO_GOALS_LINE
                        : c->kind == DGPPO_ENV_MPE_LINE ? (c->n_agents > 3 ? DGPPO_GOALS_LINE : DGPPO_GOALS_LINE_INTERIOR)
                        : c->kind == DGPPO_ENV_MPE_FORMATION ? DGPPO_GOALS_CIRCLE : DGPPO_GOALS_NODES;
    DGPPO_REQUIRE(c->reward_goals == want_rg, "reward_goals must be %d for env kind %d (got %d)", want_rg, c->kind, c->reward_goals);
    DGPPO_REQUIRE(c->n_cost == (c->kind == DGPPO_ENV_MPE_CONNECT_SPREAD ? 3 : 2), "n_cost %d does not match env kind %d",
                  c->n_cost, c->kind);
    DGPPO_REQUIRE(!(line && c->kind == DGPPO_ENV_LIDAR_LINE && c->n_agents < 2), "LidarLine needs at least 2 agents");
    DGPPO_REQUIRE(c->y_limit >= c->area_size && c->obs_mask_radius > 0.0f, "y_limit / obs_mask_radius not initialised");
  }
  DGPPO_REQUIRE(c->n_obs >= 0 && c->n_obs <= 64, "n_obs must be in [0,64] (got %d)", c->n_obs);
  const bool bicycle = cfg_is_bicycle(*c);
  DGPPO_REQUIRE(c->state_dim == (bicycle ? 5 : 4), "state_dim %d does not match env kind %d", c->state_dim, c->kind);
  DGPPO_REQUIRE(c->node_dim == c->state_dim + 3, "node_dim must be state_dim + 3");
  if (cfg_is_lidar(*c)) {
    DGPPO_REQUIRE(c->n_rays >= 1 && c->n_rays <= 256, "n_rays must be in [1,256] (got %d)", c->n_rays);
    DGPPO_REQUIRE(c->top_k >= 1 && c->top_k <= c->n_rays, "top_k must be in [1,n_rays] (got %d)", c->top_k);
  }
  return 0;
}

extern "C" int32_t dgppo_env_num_nodes(const dgppo_env_cfg* cfg) {
  if (dgppo_validate_cfg(cfg)) return -1;
  return cfg_num_nodes(*cfg);
}
extern "C" int32_t dgppo_env_num_edges(const dgppo_env_cfg* cfg) {
  if (dgppo_validate_cfg(cfg)) return -1;
  return cfg_num_edges(*cfg);
}

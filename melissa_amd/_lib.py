"""ctypes binding of libmelissa_hip.so (include/melissa_hip.h).  There is no CPU fallback: if the
library is missing the product path raises, loudly."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

MAX_HEAD_LAYERS = 6
MAX_NODES = 128              # MEL_MAX_NODES: one wavefront per graph, one or two nodes per lane, node sets = 1 or 2 uint64 words


def set_words(n: int) -> int:
    """MEL_SET_WORDS: uint64 words per node set (1 up to 64 nodes, 2 beyond)."""
    return (int(n) + 63) // 64


def check_n_nodes(n: int, what: str):
    """The kernels hold one graph per wavefront (a lane holds node i and, beyond 64 nodes, node i + 64; node sets are one
    or two 64-bit words), so graphs have at most 128 nodes: the reference CLI's --n-agents 20 / 50 / 100 (common.py:49) all
    fit."""
    if not (1 <= int(n) <= MAX_NODES):
        raise ValueError(f"{what}: {n} nodes per graph is outside [1, {MAX_NODES}] - the MI355X kernels hold one graph per "
                         f"wavefront (at most two nodes per lane, node sets of at most two 64-bit words)")
MODEL_LDGN, MODEL_HLDGN, MODEL_DGNR = 0, 1, 2
CONV_GATV2, CONV_TRANSFORMER = 0, 1
AGG = {"max": 0, "mean": 1, "add": 2}
OK, ERR_INVALID_ARG, ERR_SHAPE, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_LAUNCH = 0, -1, -2, -3, -4, -5
ENV_SCALARS, ENV_LOGGER_STATS = 16, 10
# scalars[b][k] / node_sets[b][k] / sel_sets[b][k] indices (melissa_hip.h MEL_S_* / MEL_SET_* / MEL_SEL_*)
S_ORIGIN, S_SELECTION, S_SKIP, S_NUM_MOVES, S_WORLD_MSGS, S_NEW_ROUND, S_EPISODE, S_MOVE_CURSOR, \
    S_DECISIONS, S_DONE_COUNT, S_EPISODES_DONE, S_ERROR, S_EP_CURSOR = range(13)
SET_HAS_MESSAGE, SET_ORIGIN, SET_INTERESTED, SET_SCRIPTED, SET_TRUNCATED, SET_ALIVE, SET_TERMINATED, \
    SET_AGENTS = range(8)

# every symbol include/melissa_hip.h declares
EXPORTS = ("mel_wait_counter", "mel_feature_tables_bytes", "mel_prepare_feature_tables", "mel_prepared_weights_bytes", "mel_prepare_weights", "mel_transpose_f32", "mel_episode_refill", "mel_abi_sizeof", "mel_radius_graph", "mel_gat_forward", "mel_gat_backward", "mel_pool_forward", "mel_pool_backward",
           "mel_gemm_bf16", "mel_convert_bf16", "mel_hldgn_forward_envs", "mel_hldgn_forward_envs_select", "mel_plan_pointers", "mel_select_action_envs", "mel_dgnr_forward", "mel_dgnr_forward_agents", "mel_gemm_f32", "mel_gemm_f32_t", "mel_gemm_f32_splitk", "mel_gemm_f32_split", "mel_replay_sample", "mel_adam_step", "mel_workspace_bytes", "mel_workspace_bytes_agents", "mel_ldgn_forward_agents",
           "mel_select_action_rows", "mel_ldgn_forward", "mel_hldgn_forward", "mel_forward_tap",
           "mel_select_action", "mel_env_state_bytes", "mel_env_bind", "mel_env_reset", "mel_env_step",
           "mel_env_observe", "mel_env_round", "mel_prof_create", "mel_prof_destroy", "mel_prof_attach", "mel_prof_reset",
           "mel_prof_read", "mel_last_error", "mel_version")
PREC_F32, PREC_BF16, PREC_F32_SPLIT, PREC_F32_AUTO = 0, 1, 2, 3
FWD_PLAN_READY = 1          # mel_weights.flags: the plan masks of this call were written by mel_env_round
FWD_INTEGER_FEATURES = 2    # mel_weights.flags: node features are the env's integers -> node-feature table (melissa_hip.h)
HEURISTICS = {None: 0, "simple_broadcast": 1, "broadcast_if_any_interested": 2, "silent": 3}
LOGGER_KEYS = ("total_messages_transmitted", "coverage", "messages_sent", "messages_received", "n_neighbours",
               "interested_agents", "coverage_interested_fraction", "coverage_interested_count",
               "uninterested_with_message", "episode_rewards_sum")      # graph.py:166-178
N_STAGES = 14
STAGE_NAMES = ("plan", "encoder", "conv1_lin", "conv1_lin_r", "conv1_att", "conv2_lin", "conv2_lin_r", "conv2_att",
               "head_hidden", "head_tail", "select", "env_step", "env_reset", "env_observe")


class MelLinear(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("in_dim", C.c_int32), ("out_dim", C.c_int32)]


class MelGatv2(C.Structure):
    _fields_ = [("lin_l", MelLinear), ("lin_r", MelLinear), ("att", C.c_void_p), ("bias", C.c_void_p),
                ("heads", C.c_int32), ("channels", C.c_int32), ("lin_v", MelLinear), ("kind", C.c_int32),
                ("reserved", C.c_int32)]


class MelMlp(C.Structure):
    _fields_ = [("layer", MelLinear * MAX_HEAD_LAYERS), ("n_layers", C.c_int32)]


class MelWeights(C.Structure):
    _fields_ = [("model", C.c_int32), ("in_dim", C.c_int32), ("n_actions", C.c_int32), ("dueling", C.c_int32),
                ("encoder", MelMlp), ("conv1", MelGatv2), ("conv2", MelGatv2), ("q_head", MelMlp),
                ("v_head", MelMlp), ("precision", C.c_int32), ("flags", C.c_int32), ("prepared", C.c_void_p),
                ("tables", C.c_void_p), ("tables_nodes", C.c_int32), ("reserved", C.c_int32)]


class MelSelect(C.Structure):
    _fields_ = [("act", C.c_void_p), ("eps", C.c_float), ("seed", C.c_uint32), ("step_dev", C.c_void_p),
                ("live", C.c_void_p), ("n_nodes", C.c_int32), ("reserved", C.c_int32)]


class MelRoundReplay(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("reserved", C.c_int32), ("obs", C.c_void_p), ("obs_next", C.c_void_p),
                ("acted", C.c_void_p), ("done", C.c_void_p), ("act", C.c_void_p), ("rew", C.c_void_p),
                ("episode", C.c_void_p), ("cursor", C.c_void_p)]


class MelReplayBatch(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("boot_obs", C.c_void_p), ("act", C.c_void_p), ("ret", C.c_void_p), ("boot_w", C.c_void_p),
                ("env", C.c_void_p), ("slot", C.c_void_p), ("agent", C.c_void_p)]


ADAM_MAX_TENSORS = 64


class MelAdamTensors(C.Structure):
    _fields_ = [("count", C.c_int32), ("reserved", C.c_int32), ("param", C.c_void_p * ADAM_MAX_TENSORS),
                ("grad", C.c_void_p * ADAM_MAX_TENSORS), ("exp_avg", C.c_void_p * ADAM_MAX_TENSORS),
                ("exp_avg_sq", C.c_void_p * ADAM_MAX_TENSORS), ("step", C.c_void_p * ADAM_MAX_TENSORS),
                ("numel", C.c_int64 * ADAM_MAX_TENSORS)]


class MelEnvBatch(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("n_nodes", C.c_int32), ("dynamic_graph", C.c_int32),
                ("has_local_ratio", C.c_int32), ("local_ratio", C.c_double),
                ("heuristic", C.c_int32), ("is_testing", C.c_int32),
                ("pos", C.c_void_p), ("one_hop", C.c_void_p), ("two_hop", C.c_void_p),
                ("node_sets", C.c_void_p), ("sel_sets", C.c_void_p), ("scalars", C.c_void_p),
                ("agent_msgs", C.c_void_p), ("received", C.c_void_p), ("two_hop_cover", C.c_void_p),
                ("agent_action", C.c_void_p), ("current_actions", C.c_void_p), ("steps_taken", C.c_void_p),
                ("sel_steps", C.c_void_p), ("rewards", C.c_void_p), ("pz_rewards", C.c_void_p),
                ("episode_rewards", C.c_void_p), ("obs_matrix", C.c_void_p), ("info_stats", C.c_void_p),
                ("log_capacity", C.c_int32), ("log_reserved", C.c_int32), ("log_cursor", C.c_void_p),
                ("log_stats", C.c_void_p), ("log_meta", C.c_void_p),
                ("plan_adj", C.c_void_p), ("plan_live", C.c_void_p), ("plan_u1", C.c_void_p), ("plan_u2", C.c_void_p),
                ("plan_cnt", C.c_void_p)]


class MelEpisodePool(C.Structure):
    _fields_ = [("n_episodes", C.c_int32), ("n_nodes", C.c_int32), ("max_moves", C.c_int32),
                ("reserved", C.c_int32), ("pos", C.c_void_p), ("one_hop", C.c_void_p),
                ("interested", C.c_void_p), ("origin", C.c_void_p), ("moves", C.c_void_p),
                ("scripted", C.c_void_p), ("snapshot", C.c_void_p), ("produced", C.c_void_p)]


class MelGraphPool(C.Structure):
    _fields_ = [("n_graphs", C.c_int32), ("n_nodes", C.c_int32), ("pos", C.c_void_p), ("one_hop", C.c_void_p)]


class MelEpisodeStream(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("ring", C.c_int32), ("fixed_graph", C.c_int32), ("has_density", C.c_int32),
                ("fixed_interest_density", C.c_double), ("pcg", C.c_void_p), ("pcg_half", C.c_void_p),
                ("produced", C.c_void_p), ("draw_seed", C.c_void_p), ("draw_graph", C.c_void_p), ("work", C.c_void_p),
                ("new_count", C.c_void_p)]


ENV_ERR_MOVES_EXHAUSTED, ENV_ERR_NO_SELECTION, ENV_ERR_UNCOVERED_AGENT, ENV_ERR_EPISODE_UNDERRUN = 1, 2, 4, 8


class MelEnvObs(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("obs_stride", C.c_int64), ("agent_id", C.c_void_p),
                ("action_mask", C.c_void_p), ("rew", C.c_void_p), ("terminated", C.c_void_p),
                ("flags", C.c_void_p), ("active_nb", C.c_void_p), ("stats", C.c_void_p)]


_lib = None


def lib_path() -> str:
    return _build.LIB_PATH


def load(build_if_missing: bool = True):
    """Load the shared library (building it in-tree with hipcc if absent/stale and hipcc exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  Device pointers and streams
    # borrowed from torch tensors are only meaningful inside THAT runtime instance, so torch must be
    # loaded first: our library's libamdhip64.so.7 dependency then resolves to the copy already mapped
    # (loading in the other order leaves two runtimes in the process and every launch fails with
    # "no ROCm-capable device is detected").
    import torch  # noqa: F401
    path = lib_path()
    if build_if_missing and _build.is_stale():
        try:
            _build.build_library()
        except FileNotFoundError as e:  # no hipcc on this machine: keep a prebuilt .so if there is one
            if not os.path.exists(path):
                raise RuntimeError(f"libmelissa_hip.so is missing and could not be built: {e}") from e
        # (a compile ERROR propagates: running a stale library against changed sources would hide it)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: the HIP hot path is not built (run python -m melissa_amd.build)")
    lib = C.CDLL(path)
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    W, E, P, O = C.POINTER(MelWeights), C.POINTER(MelEnvBatch), C.POINTER(MelEpisodePool), C.POINTER(MelEnvObs)
    lib.mel_last_error.restype = C.c_char_p
    lib.mel_version.restype = C.c_char_p
    lib.mel_workspace_bytes.restype = sz
    lib.mel_workspace_bytes.argtypes = [W, i64, i32]
    lib.mel_ldgn_forward.restype = i32
    lib.mel_ldgn_forward.argtypes = [W, vp, i64, i32, i32, vp, vp, sz, vp]
    lib.mel_dgnr_forward.restype = i32
    lib.mel_dgnr_forward.argtypes = [W, vp, i64, i32, i32, vp, vp, sz, vp]
    lib.mel_dgnr_forward_agents.restype = i32
    lib.mel_dgnr_forward_agents.argtypes = [W, vp, i64, i32, i32, vp, i64, vp, vp, C.POINTER(MelSelect), vp, sz, vp]
    lib.mel_hldgn_forward.restype = i32
    lib.mel_hldgn_forward.argtypes = [W, i32, vp, i64, i32, i32, vp, vp, sz, vp]
    lib.mel_gemm_f32_t.restype = i32
    lib.mel_gemm_f32_t.argtypes = [vp, i32, i32, vp, i32, i32, vp, i32, i64, i32, i32, vp]
    lib.mel_gemm_f32.restype = i32
    lib.mel_gemm_f32.argtypes = [vp, i32, vp, vp, vp, i32, i64, i32, i32, i32, i32, vp]
    lib.mel_gemm_f32_split.restype = i32
    lib.mel_gemm_f32_split.argtypes = [vp, i32, vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, vp, i64, vp]
    lib.mel_gemm_f32_splitk.restype = i32
    lib.mel_gemm_f32_splitk.argtypes = [vp, i32, vp, vp, vp, i32, i64, i32, i32, i32, i32, vp, i64, vp]
    lib.mel_gemm_bf16.restype = i32
    lib.mel_gemm_bf16.argtypes = [vp, i32, vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, vp]
    lib.mel_convert_bf16.restype = i32
    lib.mel_convert_bf16.argtypes = [vp, vp, i64, vp]
    lib.mel_abi_sizeof.restype = C.c_size_t
    lib.mel_abi_sizeof.argtypes = [i32]
    lib.mel_radius_graph.restype = i32
    lib.mel_radius_graph.argtypes = [vp, i64, i32, i32, i32, vp, vp]
    lib.mel_gat_forward.restype = i32
    lib.mel_gat_forward.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp]
    lib.mel_gat_backward.restype = i32
    lib.mel_gat_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.mel_pool_forward.restype = i32
    lib.mel_pool_forward.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, vp]
    lib.mel_pool_backward.restype = i32
    lib.mel_pool_backward.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, vp]
    lib.mel_hldgn_forward_envs.restype = i32
    lib.mel_hldgn_forward_envs.argtypes = [W, i32, vp, i64, i32, i32, vp, vp, sz, vp]
    lib.mel_plan_pointers.restype = i32
    lib.mel_plan_pointers.argtypes = [W, i64, i32, i64, vp, C.POINTER(C.c_void_p)]
    lib.mel_hldgn_forward_envs_select.restype = i32
    lib.mel_hldgn_forward_envs_select.argtypes = [W, i32, vp, i64, i32, i32, vp, C.POINTER(MelSelect), vp, sz, vp]
    lib.mel_select_action_envs.restype = i32
    lib.mel_select_action_envs.argtypes = [vp, vp, i64, i32, i32, C.c_float, C.c_uint32, vp, vp, vp]
    lib.mel_forward_tap.restype = i32
    lib.mel_forward_tap.argtypes = [W, i32, i64, i32, i64, vp, vp, vp]
    lib.mel_workspace_bytes_agents.restype = sz
    lib.mel_workspace_bytes_agents.argtypes = [W, i64, i32, i64]
    lib.mel_ldgn_forward_agents.restype = i32
    lib.mel_ldgn_forward_agents.argtypes = [W, vp, i64, i32, i32, vp, i64, vp, vp, C.POINTER(MelSelect), vp, sz, vp]
    lib.mel_select_action_rows.restype = i32
    lib.mel_select_action_rows.argtypes = [vp, vp, i64, vp, i32, C.c_float, C.c_uint32, C.c_uint32, vp, vp, vp]
    lib.mel_select_action.restype = i32
    lib.mel_select_action.argtypes = [vp, vp, i64, i32, C.c_float, vp, vp, vp, vp, vp]
    lib.mel_env_state_bytes.restype = sz
    lib.mel_env_state_bytes.argtypes = [i32, i32]
    lib.mel_env_bind.restype = i32
    lib.mel_env_bind.argtypes = [E, i32, i32, vp]
    lib.mel_env_reset.restype = i32
    lib.mel_env_reset.argtypes = [E, P, vp, vp, i64, i32, O, vp]
    lib.mel_env_step.restype = i32
    lib.mel_env_step.argtypes = [E, P, vp, vp, i64, O, vp, i32, vp]
    lib.mel_env_round.restype = i32
    lib.mel_env_round.argtypes = [E, P, vp, vp, vp, vp, i32, i32, vp, C.POINTER(MelRoundReplay), vp]
    lib.mel_adam_step.restype = i32
    lib.mel_adam_step.argtypes = [C.POINTER(MelAdamTensors), C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_double, vp]
    lib.mel_replay_sample.restype = i32
    lib.mel_replay_sample.argtypes = [C.POINTER(MelRoundReplay), i64, i32, i32, i32, C.POINTER(C.c_float), C.c_uint64, vp, vp,
                                      C.POINTER(MelReplayBatch), vp]
    lib.mel_env_observe.restype = i32
    lib.mel_env_observe.argtypes = [E, vp, i64, O, vp]
    lib.mel_feature_tables_bytes.restype = sz
    lib.mel_feature_tables_bytes.argtypes = [W, i32]
    lib.mel_prepare_feature_tables.restype = i32
    lib.mel_prepare_feature_tables.argtypes = [W, i32, vp, sz, vp]
    lib.mel_prepared_weights_bytes.restype = sz
    lib.mel_prepared_weights_bytes.argtypes = [W]
    lib.mel_prepare_weights.restype = i32
    lib.mel_prepare_weights.argtypes = [W, vp, sz, vp]
    lib.mel_transpose_f32.restype = i32
    lib.mel_transpose_f32.argtypes = [vp, i32, i64, i32, vp, i32, vp]
    lib.mel_gemm_f32.restype = i32
    lib.mel_gemm_f32.argtypes = [vp, i32, vp, vp, vp, i32, i64, i32, i32, i32, i32, vp]
    lib.mel_wait_counter.restype = i32
    lib.mel_wait_counter.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    lib.mel_episode_refill.restype = i32
    lib.mel_episode_refill.argtypes = [C.POINTER(MelEpisodeStream), C.POINTER(MelGraphPool), P, E, i32, i32, vp]
    lib.mel_prof_create.restype = vp
    lib.mel_prof_create.argtypes = [i32]
    lib.mel_prof_destroy.restype = None
    lib.mel_prof_destroy.argtypes = [vp]
    lib.mel_prof_attach.restype = None
    lib.mel_prof_attach.argtypes = [vp]
    lib.mel_prof_reset.restype = None
    lib.mel_prof_reset.argtypes = [vp]
    lib.mel_prof_read.restype = i32
    lib.mel_prof_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    _lib = lib
    return lib


def last_error() -> str:
    return load().mel_last_error().decode()


def check(status: int, what: str = ""):
    """C status -> Python exception; shape errors are ValueError like networks/common.py:20-29."""
    if status == OK:
        return
    msg = last_error()
    if status == ERR_SHAPE:
        raise ValueError(msg)
    raise RuntimeError(f"{what or 'libmelissa_hip'} failed ({status}): {msg}")


def current_stream_ptr(device=None) -> int:
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)

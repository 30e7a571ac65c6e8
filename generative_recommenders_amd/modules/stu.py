"""Drop-in for generative_recommenders/modules/stu.py: ``STULayerConfig`` (:64-80),
``STULayer`` (:175-418) with its KV cache, ``STUStack`` (:421-466) -- the caller of the op
layer.  Parameter names / shapes / initialisation match the reference so its state_dicts
load unchanged: ``_uvqk_weight, _uvqk_beta, _input_norm_weight,
_input_norm_bias, _output_weight, _output_norm_weight, _output_norm_bias``.
"""

import abc
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
from torch.autograd.profiler import record_function

from generative_recommenders_amd.common import HammerModule
from generative_recommenders_amd.ops import _launch
from generative_recommenders_amd.ops.hstu_attention import delta_hstu_mha
from generative_recommenders_amd.ops.hstu_compute import (
    hstu_fused_layer,
    hstu_fused_layer_applicable,
    hstu_compute_output,
    hstu_compute_uqvk,
    hstu_preprocess_and_attention,
)
from generative_recommenders_amd.ops.jagged_tensors import (
    asynchronous_complete_cumsum,
    concat_2D_jagged,
    split_2D_jagged,
)


class STU(HammerModule, abc.ABC):
    def cached_forward(self, delta_x: torch.Tensor, num_targets: torch.Tensor, max_kv_caching_len: int = 0,
                       kv_caching_lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError

    @abc.abstractmethod
    def forward(self, x: torch.Tensor, x_lengths: torch.Tensor, x_offsets: torch.Tensor, max_seq_len: int,
                num_targets: torch.Tensor, max_kv_caching_len: int = 0,
                kv_caching_lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        pass


@dataclass
class STULayerConfig:
    embedding_dim: int
    num_heads: int
    hidden_dim: int
    attention_dim: int
    output_dropout_ratio: float = 0.3
    causal: bool = True
    target_aware: bool = True
    max_attn_len: Optional[int] = None
    attn_alpha: Optional[float] = None
    use_group_norm: bool = False
    recompute_normed_x: bool = True
    recompute_uvqk: bool = True
    recompute_y: bool = True
    sort_by_length: bool = True
    contextual_seq_len: int = 0


def _split_cache(max_seq_len, seq_offsets, kv, kv_caching_offsets, delta_offsets):
    cache, _ = split_2D_jagged(
        max_seq_len=max_seq_len, values=kv.flatten(1, 2), max_len_left=None, max_len_right=None,
        offsets_left=kv_caching_offsets, offsets_right=delta_offsets,
    )
    return cache


class STULayer(STU):
    def __init__(self, config: STULayerConfig, is_inference: bool = False) -> None:
        super().__init__(is_inference=is_inference)
        self.reset_kv_cache()
        self._num_heads = config.num_heads
        self._embedding_dim = config.embedding_dim
        self._hidden_dim = config.hidden_dim
        self._attention_dim = config.attention_dim
        self._output_dropout_ratio = config.output_dropout_ratio
        self._target_aware = config.target_aware
        self._causal = config.causal
        self._max_attn_len = config.max_attn_len or 0
        self._attn_alpha = config.attn_alpha or 1.0 / (self._attention_dim**0.5)
        self._use_group_norm = config.use_group_norm
        self._recompute_normed_x = config.recompute_normed_x
        self._recompute_uvqk = config.recompute_uvqk
        self._recompute_y = config.recompute_y
        # MI355X: run forward() as ONE autograd node when no K/V goes to the cache (ops/hstu_compute.py::hstu_fused_layer);
        # False (or HSTU_FUSE_LAYER=0) keeps the reference's two nodes -- same results
        self.fuse_layer = os.environ.get("HSTU_FUSE_LAYER", "1") != "0"
        self._sort_by_length = config.sort_by_length
        self._contextual_seq_len = config.contextual_seq_len

        proj = (self._hidden_dim * 2 + self._attention_dim * 2) * self._num_heads
        self._uvqk_weight = torch.nn.Parameter(torch.empty((self._embedding_dim, proj)))
        torch.nn.init.xavier_uniform_(self._uvqk_weight)
        self._uvqk_beta = torch.nn.Parameter(torch.zeros((proj,)))
        self._input_norm_weight = torch.nn.Parameter(torch.ones((self._embedding_dim,)))
        self._input_norm_bias = torch.nn.Parameter(torch.zeros((self._embedding_dim,)))
        self._output_weight = torch.nn.Parameter(
            torch.empty((self._hidden_dim * self._num_heads * 3, self._embedding_dim)))
        torch.nn.init.xavier_uniform_(self._output_weight)
        norm_shape = self._num_heads if self._use_group_norm else self._hidden_dim * self._num_heads
        self._output_norm_weight = torch.nn.Parameter(torch.ones((norm_shape,)))
        self._output_norm_bias = torch.nn.Parameter(torch.zeros((norm_shape,)))

    # ---- KV cache (stu.py:84-172, 247-289) ----
    def reset_kv_cache(self) -> None:
        self.k_cache: Optional[torch.Tensor] = None
        self.v_cache: Optional[torch.Tensor] = None
        self.kv_caching_offsets: Optional[torch.Tensor] = None
        self.max_kv_caching_len: int = 0
        # persistent [cache ; delta] buffers of the in-place append path (see construct_full_kv)
        self._kv_full: Optional[Tuple[torch.Tensor, torch.Tensor, tuple, torch.Tensor]] = None

    def update_kv_cache(self, max_seq_len: int, seq_offsets: torch.Tensor, k: Optional[torch.Tensor],
                        v: Optional[torch.Tensor], max_kv_caching_len: int,
                        kv_caching_lengths: Optional[torch.Tensor]) -> None:
        if kv_caching_lengths is None:
            return
        kv_caching_offsets = asynchronous_complete_cumsum(kv_caching_lengths)
        delta_offsets = seq_offsets - kv_caching_offsets
        self.k_cache = _split_cache(max_seq_len, seq_offsets, k, kv_caching_offsets, delta_offsets)
        self.v_cache = _split_cache(max_seq_len, seq_offsets, v, kv_caching_offsets, delta_offsets)
        if max_kv_caching_len == 0:
            max_kv_caching_len = int(kv_caching_lengths.max().item())
        self.max_kv_caching_len = max_kv_caching_len
        self.kv_caching_offsets = kv_caching_offsets
        self._kv_full = None                      # the cache changed: the [cache ; delta] buffers are stale

    def construct_full_kv(self, delta_k: torch.Tensor, delta_v: torch.Tensor
                          ) -> Tuple[torch.Tensor, torch.Tensor, int, torch.Tensor]:
        """[cached rows ; delta rows] per user as one jagged tensor + its offsets (stu.py:134-172).

        The reference rebuilds both tensors with ``concat_2D_jagged`` on every call: O(history) bytes per M-FALCON
        microbatch and layer.  Under ``torch.no_grad()`` (inference: nothing holds on to earlier results) the buffers of
        the first call are kept and later calls with the same microbatch size overwrite only the delta rows in place
        (``hstu_jagged_write_tail``): O(delta).  The returned k / v are then views of those buffers -- valid until the
        next ``cached_forward`` / ``update_kv_cache`` of this layer, which is all ``cached_forward`` needs."""
        L, _ = delta_k.shape
        B = self.kv_caching_offsets.shape[0] - 1
        delta_size = L // B
        in_place = not torch.is_grad_enabled() and not (delta_k.requires_grad or delta_v.requires_grad)
        # the buffers are reused only for the very cache tensors they were built from: k_cache / v_cache /
        # kv_caching_offsets are public attributes (a caller may assign them directly, as with the reference), so the
        # key holds their storage pointers, the device and the dtype next to the microbatch size
        key = (delta_size, delta_k.dtype, delta_k.device, self.k_cache.data_ptr(), self.v_cache.data_ptr(),
               self.kv_caching_offsets.data_ptr(), self.max_kv_caching_len)
        if in_place and self._kv_full is not None and self._kv_full[2] == key:
            k_full, v_full, _, full_offsets = self._kv_full
            _launch.jagged_write_tail_(k_full, delta_k, full_offsets, delta_size)
            _launch.jagged_write_tail_(v_full, delta_v, full_offsets, delta_size)
            return k_full, v_full, self.max_kv_caching_len + delta_size, full_offsets
        full = []
        for cache, delta in ((self.k_cache, delta_k), (self.v_cache, delta_v)):
            full.append(concat_2D_jagged(
                max_seq_len=self.max_kv_caching_len + delta_size, values_left=cache, values_right=delta,
                max_len_left=self.max_kv_caching_len, max_len_right=delta_size,
                offsets_left=self.kv_caching_offsets, offsets_right=None))
        full_offsets = self.kv_caching_offsets + delta_size * torch.arange(
            B + 1, device=delta_k.device, dtype=self.kv_caching_offsets.dtype)
        if in_place:
            self._kv_full = (full[0], full[1], key, full_offsets)
        return full[0], full[1], self.max_kv_caching_len + delta_size, full_offsets

    # ---- forward (stu.py:291-352) ----
    def forward(self, x: torch.Tensor, x_lengths: torch.Tensor, x_offsets: torch.Tensor, max_seq_len: int,
                num_targets: torch.Tensor, max_kv_caching_len: int = 0,
                kv_caching_lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        if (self.fuse_layer and kv_caching_lengths is None and self._causal
                and hstu_fused_layer_applicable(x, self._attention_dim, self._hidden_dim)):
            # nothing to hand to the K/V cache: the whole layer as one autograd node (SiLU and the residual's gradient
            # folded into the neighbouring row kernels; identical results)
            with record_function("## stu_layer ##"):
                out = hstu_fused_layer(
                    x=x,
                    input_norm_weight=self._input_norm_weight, input_norm_bias=self._input_norm_bias, input_norm_eps=1e-6,
                    uvqk_weight=self._uvqk_weight, uvqk_bias=self._uvqk_beta,
                    output_norm_weight=self._output_norm_weight, output_norm_bias=self._output_norm_bias,
                    output_norm_eps=1e-6, output_weight=self._output_weight,
                    num_heads=self._num_heads, attn_dim=self._attention_dim, hidden_dim=self._hidden_dim,
                    max_seq_len=max_seq_len, seq_offsets=x_offsets, attn_alpha=self._attn_alpha,
                    num_targets=num_targets if self._target_aware else None,
                    max_attn_len=self._max_attn_len, contextual_seq_len=self._contextual_seq_len,
                    dropout_ratio=self._output_dropout_ratio, training=self.training, concat_ux=True,
                    group_norm=self._use_group_norm,
                    recompute_uvqk_in_backward=self._recompute_uvqk,
                    recompute_normed_x_in_backward=self._recompute_normed_x,
                    recompute_y_in_backward=self._recompute_y, sort_by_length=self._sort_by_length)
            self.update_kv_cache(max_seq_len=max_seq_len, seq_offsets=x_offsets, k=None, v=None,
                                 max_kv_caching_len=max_kv_caching_len, kv_caching_lengths=kv_caching_lengths)
            return out
        with record_function("## stu_preprocess_and_attention ##"):
            u, attn_output, k, v = hstu_preprocess_and_attention(
                x=x,
                norm_weight=self._input_norm_weight,
                norm_bias=self._input_norm_bias,
                norm_eps=1e-6,
                num_heads=self._num_heads,
                attn_dim=self._attention_dim,
                hidden_dim=self._hidden_dim,
                uvqk_weight=self._uvqk_weight,
                uvqk_bias=self._uvqk_beta,
                max_seq_len=max_seq_len,
                seq_offsets=x_offsets,
                attn_alpha=self._attn_alpha,
                causal=self._causal,
                num_targets=num_targets if self._target_aware else None,
                max_attn_len=self._max_attn_len,
                contextual_seq_len=self._contextual_seq_len,
                recompute_uvqk_in_backward=self._recompute_uvqk,
                recompute_normed_x_in_backward=self._recompute_normed_x,
                sort_by_length=self._sort_by_length,
                prefill=kv_caching_lengths is not None,
                kernel=self.hammer_kernel(),
            )
        self.update_kv_cache(max_seq_len=max_seq_len, seq_offsets=x_offsets, k=k, v=v,
                             max_kv_caching_len=max_kv_caching_len, kv_caching_lengths=kv_caching_lengths)
        with record_function("## stu_compute_output ##"):
            return self._output(attn_output, u, x)

    def _output(self, attn: torch.Tensor, u: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        return hstu_compute_output(
            attn=attn, u=u, x=x,
            norm_weight=self._output_norm_weight,
            norm_bias=self._output_norm_bias,
            norm_eps=1e-6,
            dropout_ratio=self._output_dropout_ratio,
            output_weight=self._output_weight,
            group_norm=self._use_group_norm,
            num_heads=self._num_heads,
            linear_dim=self._hidden_dim,
            concat_ux=True,
            training=self.training,
            kernel=self.hammer_kernel(),
            recompute_y_in_backward=self._recompute_y,
        )

    # ---- incremental forward over the KV cache (stu.py:354-418) ----
    def cached_forward(self, delta_x: torch.Tensor, num_targets: torch.Tensor, max_kv_caching_len: int = 0,
                       kv_caching_lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        with record_function("## stu_compute_uqvk ##"):
            delta_u, delta_q, delta_k, delta_v = hstu_compute_uqvk(
                x=delta_x,
                norm_weight=self._input_norm_weight.to(delta_x.dtype),
                norm_bias=self._input_norm_bias.to(delta_x.dtype),
                norm_eps=1e-6,
                num_heads=self._num_heads,
                attn_dim=self._attention_dim,
                hidden_dim=self._hidden_dim,
                uvqk_weight=self._uvqk_weight.to(delta_x.dtype),
                uvqk_bias=self._uvqk_beta.to(delta_x.dtype),
                kernel=self.hammer_kernel(),
            )
        k, v, max_seq_len, seq_offsets = self.construct_full_kv(delta_k=delta_k.flatten(1, 2),
                                                                delta_v=delta_v.flatten(1, 2))
        k = k.view(-1, self._num_heads, self._attention_dim)
        v = v.view(-1, self._num_heads, self._hidden_dim)
        self.update_kv_cache(max_seq_len=max_seq_len, seq_offsets=seq_offsets, k=k, v=v,
                             max_kv_caching_len=max_kv_caching_len, kv_caching_lengths=kv_caching_lengths)
        with record_function("## delta_hstu_mha ##"):
            delta_attn = delta_hstu_mha(
                max_seq_len=max_seq_len, alpha=self._attn_alpha, delta_q=delta_q, k=k, v=v,
                seq_offsets=seq_offsets, num_targets=num_targets if self._target_aware else None,
                max_attn_len=self._max_attn_len, contextual_seq_len=self._contextual_seq_len,
                kernel=self.hammer_kernel(),
            ).reshape(-1, self._hidden_dim * self._num_heads)
        with record_function("## stu_compute_output ##"):
            return self._output(delta_attn, delta_u, delta_x)


class STUStack(STU):
    def __init__(self, stu_list: List[STU], is_inference: bool = False) -> None:
        super().__init__(is_inference=is_inference)
        self._stu_layers = torch.nn.ModuleList(modules=stu_list)

    def forward(self, x: torch.Tensor, x_lengths: torch.Tensor, x_offsets: torch.Tensor, max_seq_len: int,
                num_targets: torch.Tensor, max_kv_caching_len: int = 0,
                kv_caching_lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        for layer in self._stu_layers:
            x = layer(x=x, x_lengths=x_lengths, x_offsets=x_offsets, max_seq_len=max_seq_len,
                      num_targets=num_targets, max_kv_caching_len=max_kv_caching_len,
                      kv_caching_lengths=kv_caching_lengths)
        return x

    def cached_forward(self, delta_x: torch.Tensor, num_targets: torch.Tensor, max_kv_caching_len: int = 0,
                       kv_caching_lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        for layer in self._stu_layers:
            delta_x = layer.cached_forward(delta_x=delta_x, num_targets=num_targets,
                                           max_kv_caching_len=max_kv_caching_len,
                                           kv_caching_lengths=kv_caching_lengths)
        return delta_x

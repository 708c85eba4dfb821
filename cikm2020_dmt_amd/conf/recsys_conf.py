"""INI configuration of the trainer, mirroring /root/reference/DMT_code/conf/recsys_conf.py:17-365 (class Conf).

Same section / key names (util/util.py:5-113 constants), same typed coercions for the keys the hot path reads, the
embedding mini-language `Name:size:dim:feature:side#...` (recsys_conf.py:274-284) and the attention pairs
`seq:item#...|...` (:295-305).  Path handling, e-mail / online-learning keys and HDFS helpers are out of scope
(SURVEY.md §2); `to_spec()` yields the plain dict the engine consumes.
"""
from __future__ import annotations

import configparser
import os

INFO, TYPE = "info", "type"
PARAMETER, LOSS_WEIGHT, LOSS_WEIGHT_METHOD = "parameter", "loss_weight", "loss_weight_method"
MODEL, MODEL_TYPE, FEAT_DIM, OUTPUT_UNITS = "model", "model_type", "feature_dimension", "output_units"
HIDDEN_UNITS_BIAS, hidden_units_bottom, hidden_units_task, num_experts = "hidden_units_bias", "hidden_units_bottom", "hidden_units_task", "num_experts"
loss_unbias_method, LOSS_CTR_REL_METHOD, dropout_rate_bias = "loss_unbias_method", "loss_ctr_rel_method", "dropout_rate_bias"
BATCH_SIZE, EPOCH_NUM, LEARNING_RATE, STEP_BOUNDARY, OPTIMIZER = "batch_size", "epoch_num", "learning_rate", "step_boundary", "optimizer"
GPU_VISIBLE, IS_BN, IS_DROPOUT, WND_WD, zero_pad, IS_USE_FEATURE = "gpu_visible", "is_bn", "is_dropout", "wnd_wd", "zero_pad", "is_use_feature"
CLASS_WEIGHT, TRAIN_WEIGHT, VALID_WEIGHT, WEIGHT_CTR, WEIGHT_ECVR = "class_weight", "train_weight", "valid_weight", "weight_ctr", "weight_ecvr"
EMBEDDING, EMB, EMB_BIAS, ATTENTION_EMBED, attention_embed_seq_ts = "embedding", "emb", "emb_bias", "attention_embed", "attention_embed_seq_ts"
SCHEMA, HEADER_SCHEMA = "schema", "header_schema"


def str_to_bool(s):
    return s in ["True", "true", "yes", "TRUE", "1"]


def csv_to_int_list(s):
    return [int(a) for a in s.strip().split(",")]


def csv_to_float_list(s):
    return [float(a) for a in s.strip().split(",")]


def parse_weight(s):
    d = {}
    for kv in s.split(","):
        k, v = kv.split(":")
        d[int(k)] = float(v)
    return [d[k] for k in sorted(d)]


class Conf:
    def __init__(self, conf_path="./", conf_file="dmt.conf"):
        self.conf_parser = configparser.ConfigParser()
        full = os.path.join(conf_path, conf_file)
        if not self.conf_parser.read(full):
            raise IOError("cannot read configuration %s" % full)
        self.tag = conf_file[:-5] if conf_file.endswith(".conf") else conf_file
        self.conf_sections = {sec: dict(self.conf_parser.items(sec)) for sec in self.conf_parser.sections()}
        self.reset(PARAMETER, LOSS_WEIGHT, csv_to_float_list, None)
        for key, f, default in ((FEAT_DIM, int, None), (OUTPUT_UNITS, int, None), (HIDDEN_UNITS_BIAS, csv_to_int_list, None),
                                (hidden_units_bottom, csv_to_int_list, None), (hidden_units_task, csv_to_int_list, None),
                                (num_experts, int, None), (IS_USE_FEATURE, str_to_bool, True), (EPOCH_NUM, int, None),
                                (BATCH_SIZE, int, None), (IS_BN, str_to_bool, None), (IS_DROPOUT, str_to_bool, None),
                                (WND_WD, float, None), (LOSS_CTR_REL_METHOD, str, None), (dropout_rate_bias, csv_to_float_list, None)):
            self.reset(MODEL, key, f, default)
        for key in (TRAIN_WEIGHT, VALID_WEIGHT, WEIGHT_CTR, WEIGHT_ECVR):
            self.reset(CLASS_WEIGHT, key, parse_weight, None)
        self[MODEL][STEP_BOUNDARY] = [int(x) for x in self[MODEL][STEP_BOUNDARY].split(",")]
        self[MODEL][LEARNING_RATE] = [float(x) for x in self[MODEL][LEARNING_RATE].split(",")]
        self.embedding_list = self.get_emb(self[EMBEDDING][EMB])
        self.embedding_list_bias = self.get_emb(self[EMBEDDING].get(EMB_BIAS, ""))
        self.attention_embed_pairs = self.get_attention_embed_v2(self[EMBEDDING][ATTENTION_EMBED])
        self.attention_embed_seq_ts = self.get_attention_embed_ts(self[EMBEDDING].get(attention_embed_seq_ts, ""))
        self.weight_ctr = self[CLASS_WEIGHT][WEIGHT_CTR]
        self.weight_ecvr = self[CLASS_WEIGHT][WEIGHT_ECVR]
        if SCHEMA in self.conf_sections:
            self[SCHEMA][HEADER_SCHEMA] = [s.strip() for s in self[SCHEMA][HEADER_SCHEMA].split(",")]
        self.model_type = self[MODEL][MODEL_TYPE]
        self.zero_pad = self[MODEL].get(zero_pad, "true")        # raw string, always truthy
        self.is_unbias_model = "unbias" in self.model_type
        if self.is_unbias_model:
            self.loss_unbias_method = self[MODEL][loss_unbias_method]
            self.dropout_rate_bias = self[MODEL][dropout_rate_bias]
            self.loss_ctr_rel_method = self[MODEL][LOSS_CTR_REL_METHOD]
        self.is_use_feature = self[MODEL][IS_USE_FEATURE]
        if "transformer" in self.model_type:
            m = self[MODEL]
            self.d_model = int(m["transformer_d_model"])
            self.d_ff = int(m["transformer_d_ff"])
            self.num_heads = int(m["transformer_num_heads"])
            self.num_blocks_encode = int(m["transformer_num_blocks_encode"])
            self.num_blocks_decode = int(m["transformer_num_blocks_decode"])
            self.maxlen_k = int(m["transformer_maxlen_k"])
            self.maxlen_q = int(m["transformer_maxlen_q"])
            self.dropout_rate = float(m["transformer_dropout_rate"])
            self.is_trans_input_by_mlp = str_to_bool(m.get("transformer_is_trans_input_by_mlp", "false"))
            self.position_encoding_method = m.get("transformer_position_encoding_method", "position_sin_cos")
            self.is_use_seq_ts = len(self[EMBEDDING].get(attention_embed_seq_ts, "")) >= 1
            self.is_trans_out_concat_item = str_to_bool(m.get("transformer_is_trans_out_concat_item", "true"))
            self.is_trans_out_by_mlp = str_to_bool(m.get("transformer_is_trans_out_by_mlp", "false"))
            self.is_decoder_add_pos_emb = str_to_bool(m.get("transformer_is_decoder_add_pos_emb", "false"))

    def reset(self, section, option, f, default):
        try:
            self.conf_sections[section][option] = f(self.conf_sections[section][option])
        except Exception:
            self.conf_sections.setdefault(section, {})[option] = default

    def __getitem__(self, k):
        return self.conf_sections[k]

    @staticmethod
    def get_emb(embs):
        if len(embs) <= 2:
            return []
        out = []
        for e in embs.split("#"):
            f = e.split(":")
            f[1], f[2] = int(f[1]), int(f[2])
            out.append(f)
        return out

    @staticmethod
    def get_attention_embed_v2(attention):
        if len(attention) <= 2:
            return []
        return [[tuple(p.split(":")[:2]) for p in grp.split("#")] for grp in attention.split("|")]

    @staticmethod
    def get_attention_embed_ts(attention):
        if len(attention) <= 1:
            return []
        return [a.strip() for a in attention.split("|")]

    def get_idschema(self):
        return [e[3] for e in self.embedding_list]

    def get_idschema_bias(self):
        return [e[3] for e in self.embedding_list_bias]

    def to_spec(self) -> dict:
        """Plain-dict model spec for cikm2020_dmt_amd.engine (only the default DMT options are supported)."""
        if self.model_type not in ("mmoe_transformer_unbias", "mmoe_transformer"):
            raise NotImplementedError("model_type %s is outside the DMT hot path" % self.model_type)
        if self.position_encoding_method not in ("position_learn", "position_sin_cos") or self.num_blocks_encode < 1 or self.num_blocks_decode < 1:
            raise NotImplementedError("Transformer options outside the engine: position_learn / position_sin_cos, is_decoder_add_pos_emb, "
                                      "is_trans_input_by_mlp, is_trans_out_concat_item (with or without is_trans_out_by_mlp), >= 1 blocks each way are implemented "
                                      "(dmt.conf ships position_learn, everything else off)")
        if self[MODEL][IS_BN] or self[MODEL][IS_DROPOUT]:
            raise NotImplementedError("is_bn / is_dropout are false in dmt.conf and not implemented")
        if str(self[PARAMETER].get(LOSS_WEIGHT_METHOD, "fixed")).strip() != "fixed":
            # inference_mlp.py:216-219, 251-254 weights the task losses by exp(-model.click_weight) / exp(-model.order_weight); those two
            # variables ('uncertainty_click_weight' / 'uncertainty_order_weight') are created by model/net/multi_task.py:124-128 and
            # multi_task_transformer.py:181-185 ONLY -- mmoe_transformer[_unbias].py (the DMT model this path rebuilds) and base.py never
            # define them, so the reference itself fails with AttributeError for this model type.  Same outcome here, said plainly.
            raise NotImplementedError("loss_weight_method = %s: the DMT model (mmoe_transformer[_unbias]) has no uncertainty_click_weight / "
                                      "uncertainty_order_weight variables in the reference either (only multi_task*.py creates them); "
                                      "use 'fixed' (dmt.conf)" % self[PARAMETER].get(LOSS_WEIGHT_METHOD))
        m = self[MODEL]
        return dict(
            embedding_list=[tuple(e) for e in self.embedding_list], embedding_list_bias=[tuple(e) for e in self.embedding_list_bias],
            attention_embed_pairs=self.attention_embed_pairs, attention_embed_seq_ts=self.attention_embed_seq_ts,
            feature_dimension=m[FEAT_DIM], d_model=self.d_model, d_ff=self.d_ff, num_heads=self.num_heads, maxlen_k=self.maxlen_k,
            num_blocks_encode=int(self.num_blocks_encode), num_blocks_decode=int(self.num_blocks_decode), hidden_units_bottom=m[hidden_units_bottom], hidden_units_task=m[hidden_units_task],
            num_experts=m[num_experts], num_tasks=2, hidden_units_bias=m[HIDDEN_UNITS_BIAS] or [], output_units=m[OUTPUT_UNITS],
            weight_ctr=self.weight_ctr, weight_ecvr=self.weight_ecvr, loss_weight=self[PARAMETER][LOSS_WEIGHT],
            loss_unbias_method=getattr(self, "loss_unbias_method", "two_head_add"),
            loss_ctr_rel_method=getattr(self, "loss_ctr_rel_method", "ctr_rel"), tie_ffn=True, dropout_rate=self.dropout_rate,
            dropout_rate_bias=getattr(self, "dropout_rate_bias", [0.5, 0.5]), position_encoding_method=self.position_encoding_method,
            is_decoder_add_pos_emb=bool(self.is_decoder_add_pos_emb), is_trans_out_concat_item=bool(self.is_trans_out_concat_item),
            is_trans_out_by_mlp=bool(self.is_trans_out_concat_item and self.is_trans_out_by_mlp),
            is_trans_input_by_mlp=bool(self.is_trans_input_by_mlp),
        )

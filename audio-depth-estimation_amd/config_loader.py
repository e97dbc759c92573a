"""yaml -> nested SimpleNamespace config (mirror of the reference's config_loader.py:43-96).

cfg.dataset.* / cfg.mode.* / cfg.model.* come from conf/dataset/<name>.yaml, conf/mode/<mode>.yaml and
conf/model/<model>.yaml next to this file; a tiny ``key: value`` parser covers the case where PyYAML is
missing (reference :8-41).  Pure host code: nothing to accelerate here.
"""
import os
import re
from types import SimpleNamespace

_CONF = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conf')


def _scalar(text):
    text = text.split('#')[0].strip() if '#' in text else text.strip()
    low = text.lower()
    if low == 'true':
        return True
    if low == 'false':
        return False
    if low == 'null' or text == '':
        return None
    if text.isdigit():
        return int(text)
    if re.match(r'^-?\d+\.\d+$', text):
        return float(text)
    if len(text) >= 2 and text[0] == text[-1] and text[0] in '"\'':
        return text[1:-1]
    return text


def _parse_yaml_simple(filepath):
    cfg = {}
    with open(filepath, 'r') as f:
        for raw in f:
            line = raw.strip()
            if not line or line.startswith('#') or ':' not in line:
                continue
            key, value = line.split(':', 1)
            cfg[key.strip()] = _scalar(value)
    return cfg


def _read(path):
    try:
        import yaml
    except ImportError:
        return _parse_yaml_simple(path)
    with open(path, 'r') as f:
        return yaml.safe_load(f)


def load_config(dataset_name='batvisionv2', mode='train', experiment_name='default', model_name='unet_baseline'):
    """dataset_name: batvisionv1|batvisionv2; mode: train|test; model_name falls back to unet_baseline."""
    model_file = os.path.join(_CONF, 'model', f'{model_name}.yaml')
    if not os.path.exists(model_file):
        model_file = os.path.join(_CONF, 'model', 'unet_baseline.yaml')
    cfg = SimpleNamespace()
    cfg.dataset = SimpleNamespace(**_read(os.path.join(_CONF, 'dataset', f'{dataset_name}.yaml')))
    cfg.mode = SimpleNamespace(**_read(os.path.join(_CONF, 'mode', f'{mode}.yaml')))
    cfg.mode.mode = mode
    cfg.mode.experiment_name = experiment_name
    cfg.model = SimpleNamespace(**_read(model_file))
    return cfg

"""Input formats either side of the hot path: the corruption streams the
reference reads (`data/tta_datasets.py:11-36,98-273`, `data/data_utils.py:5-26`), the precomputed
text bank (`Uni_Adapter.py:298-300`) and Uni3D / ULIP checkpoints (`utils/load_models.py:53-59`).

Files are read with loaders that execute nothing from the file: `numpy.load(allow_pickle=False)`
(the reference passes allow_pickle=True; plain float/int arrays load identically) and
`torch.load(weights_only=True)`.  A dataset item is `(pointcloud (N,3) float32, label int,
class_name str, rgb (N,3) ones)` exactly as the reference's `__getitem__` returns it."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

# label id -> name, in dataset order (ModelNet40, ScanObjectNN, ShapeNetCore-55 category lists)
CLASS_NAMES = {
    'modelnet': ['airplane', 'bathtub', 'bed', 'bench', 'bookshelf', 'bottle', 'bowl', 'car', 'chair', 'cone', 'cup', 'curtain',
                 'desk', 'door', 'dresser', 'flower_pot', 'glass_box', 'guitar', 'keyboard', 'lamp', 'laptop', 'mantel', 'monitor',
                 'night_stand', 'person', 'piano', 'plant', 'radio', 'range_hood', 'sink', 'sofa', 'stairs', 'stool', 'table',
                 'tent', 'toilet', 'tv_stand', 'vase', 'wardrobe', 'xbox'],
    'scanobject': ['bag', 'bin', 'box', 'cabinet', 'chair', 'desk', 'display', 'door', 'shelf', 'table', 'bed', 'pillow', 'sink',
                   'sofa', 'toilet'],
    'shapenet': ['airplane', 'bag', 'basket', 'bathtub', 'bed', 'bench', 'bottle', 'bowl', 'bus', 'cabinet', 'can', 'camera',
                 'cap', 'car', 'chair', 'clock', 'dishwasher', 'monitor', 'table', 'telephone', 'tin_can', 'tower', 'train',
                 'keyboard', 'earphone', 'faucet', 'file', 'guitar', 'helmet', 'jar', 'knife', 'lamp', 'laptop', 'speaker',
                 'mailbox', 'microphone', 'microwave', 'motorcycle', 'mug', 'piano', 'pillow', 'pistol', 'pot', 'printer',
                 'remote_control', 'rifle', 'rocket', 'skateboard', 'sofa', 'stove', 'vessel', 'washer', 'cellphone',
                 'birdhouse', 'bookshelf'],
}

CORRUPTIONS = ['uniform', 'gaussian', 'background', 'impulse', 'upsampling', 'distortion_rbf', 'distortion_rbf_inv', 'density',
               'density_inc', 'shear', 'rotation', 'cutout', 'distortion', 'occlusion', 'lidar']      # main_test-time.py:55-59


def corruption_files(root, corruption, severity):
    """File naming of `data/tta_datasets.py:11-27`: returns (data_file, label_file)."""
    if 'mixed_corruptions' in corruption:
        return os.path.join(root, f'{corruption}.npy'), os.path.join(root, 'mixed_corruptions_labels.npy')
    if corruption == 'clean':
        return os.path.join(root, 'data_original.npy'), os.path.join(root, 'label.npy')
    return os.path.join(root, f'data_{corruption}_{severity}.npy'), os.path.join(root, 'label.npy')


def load_data(root, corruption, severity, mmap=False):
    """(data (n,N,3+), labels) of one corruption stream.  `mmap=True` maps the file instead of reading it."""
    data_file, label_file = corruption_files(root, corruption, severity)
    if not os.path.exists(data_file):
        raise FileNotFoundError(f'Data file not found: {data_file}')
    if not os.path.exists(label_file):
        raise FileNotFoundError(f'Label file not found: {label_file}')
    try:
        data = np.load(data_file, allow_pickle=False, mmap_mode='r' if mmap else None)
        label = np.load(label_file, allow_pickle=False)
    except ValueError as e:                                    # object arrays need pickle: refuse, say why
        raise ValueError(f'{data_file} / {label_file}: object-dtype .npy files are not loaded (pickle); '
                         f're-save them as plain float32 / int64 arrays') from e
    return data, label


def _dataset_key(dataset_name):
    name = dataset_name.lower()
    for key in ('modelnet', 'scanobject', 'shapenet'):
        if key in name:
            return key
    raise NotImplementedError(f'Dataset {dataset_name} is not implemented')


class CorruptionStream(Dataset):
    """ModelNet40-C / ScanObjectNN-C / ShapeNetCore-C stream (`tta_datasets.py:98-168,234-273`)."""

    def __init__(self, args, root=None):
        self.key = _dataset_key(args.dataset_name)
        self.class_name = CLASS_NAMES[self.key]
        self.data, label = load_data(root if root is not None else args.myroot, args.corruption, args.severity,
                                     mmap=getattr(args, 'mmap', False))
        label = np.asarray(label)
        # ScanObjectNN labels are sometimes stored as (1,n) or (n,1) (`:156-160`)
        self.label = label.reshape(-1).astype(np.int64)
        if len(self.label) != len(self.data):
            raise ValueError(f'{len(self.data)} clouds but {len(self.label)} labels')
        if getattr(args, 'debug', False):                       # `:107-109`
            self.data, self.label = self.data[:5], self.label[:5]

    def __len__(self):
        return len(self.data)

    def __getitem__(self, item):
        pointcloud = np.asarray(self.data[item], dtype=np.float32)
        label = int(self.label[item])
        return pointcloud, label, self.class_name[label], torch.ones(pointcloud.shape, dtype=torch.float32)   # rgb = ones `:127`


def load_tta_dataset(args):
    """`data/data_utils.py:5-26`."""
    return CorruptionStream(args)


def load_text_bank(path, device='cpu'):
    """(K,D) fp32 text features (`Uni_Adapter.py:298-300`)."""
    t = torch.load(path, map_location=device, weights_only=True)
    if not isinstance(t, torch.Tensor) or t.dim() != 2:
        raise ValueError(f'{path}: expected a (K,D) tensor, got {type(t).__name__}')
    return t.float()


def load_point_encoder(args, device='cuda'):
    """Build the MI355X encoder from a checkpoint in the reference's layout (`utils/load_models.py:53-59,64-106`):
    Uni3D: torch.save({'module': state_dict}); ULIP: {'state_dict': ...} with `module.` prefixes."""
    from .point_encoder import create_openshape_from_state_dict, create_ulip_from_state_dict, create_uni3d_from_state_dict
    vlm = getattr(args, 'vlm3d', 'uni3d')
    key = {'uni3d': 'pretrained_pc_uni3d', 'ulip': 'pretrained_pc_ulip', 'openshape': 'pretrained_pc_oshape'}.get(vlm, 'ckpt_path')
    path = getattr(args, key, None) or getattr(args, 'ckpt_path', None)
    if not path or not os.path.exists(path):
        raise FileNotFoundError(f'{vlm} checkpoint not found: {path}')
    ck = torch.load(path, map_location='cpu', weights_only=True)
    sd = ck.get('module', ck.get('state_dict', ck)) if isinstance(ck, dict) else ck
    precision = getattr(args, 'gemm_precision', 'split')
    if vlm == 'uni3d':
        return create_uni3d_from_state_dict(sd, args, device=device, precision=precision)
    if vlm == 'ulip':
        return create_ulip_from_state_dict(sd, device=device, precision=precision)
    if vlm == 'openshape':                 # the reference's loader is a stub (utils/load_models.py:108-127): the wiring is ours
        return create_openshape_from_state_dict(sd, device=device, precision=precision)
    raise NotImplementedError(f'--vlm3d {vlm}')

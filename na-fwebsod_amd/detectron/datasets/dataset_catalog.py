"""Dataset name -> paths (reference: detectron/datasets/dataset_catalog.py:28-304, the entries
the WSL configs name).  Same accessors; `register` adds datasets at run time (tests, custom data)."""
import os

_DATA_DIR = os.path.join(os.path.dirname(__file__), 'data')

_IM_DIR, _ANN_FN, _IM_PREFIX, _DEVKIT_DIR = 'image_directory', 'annotation_file', 'image_prefix', 'devkit_directory'


def _voc(year, split):
    return {_IM_DIR: _DATA_DIR + '/VOC%s/JPEGImages' % year,
            _ANN_FN: _DATA_DIR + '/VOC%s/annotations/voc_%s_%s.json' % (year, year, split),
            _DEVKIT_DIR: _DATA_DIR + '/VOC%s/VOCdevkit%s' % (year, year)}


_DATASETS = {
    'flickr_voc': {_IM_DIR: _DATA_DIR + '/flickr_voc/images', _ANN_FN: _DATA_DIR + '/flickr_voc/images.json',
                   _DEVKIT_DIR: _DATA_DIR + '/VOC2007/VOCdevkit2007'},
    'flickr_coco': {_IM_DIR: _DATA_DIR + '/flickr_coco/images',
                    _ANN_FN: _DATA_DIR + '/flickr_coco/images.json',
                    _DEVKIT_DIR: _DATA_DIR + '/VOC2007/VOCdevkit2007'},
    # (upstream spells these two 'image', not 'images': dataset_catalog.py:253-260)
    'flickr_clean': {_IM_DIR: _DATA_DIR + '/flickr_clean/image',
                     _ANN_FN: _DATA_DIR + '/flickr_clean/image.json',
                     _DEVKIT_DIR: _DATA_DIR + '/VOC2007/VOCdevkit2007'},
    'coco_2014_train': {_IM_DIR: _DATA_DIR + '/coco/coco_train2014',
                        _ANN_FN: _DATA_DIR + '/coco/annotations/instances_train2014.json'},
    'coco_2014_val': {_IM_DIR: _DATA_DIR + '/coco/coco_val2014',
                      _ANN_FN: _DATA_DIR + '/coco/annotations/instances_val2014.json'},
    'coco_2014_minival': {_IM_DIR: _DATA_DIR + '/coco/coco_val2014',
                          _ANN_FN: _DATA_DIR + '/coco/annotations/instances_minival2014.json'},
}
for _y in ('2007', '2012'):
    for _s in ('train', 'val', 'trainval', 'test'):
        _DATASETS['voc_%s_%s' % (_y, _s)] = _voc(_y, _s)


_REGISTERED = {}


def register(name, image_directory, annotation_file, image_prefix='', devkit_directory=None):
    _DATASETS[name] = {_IM_DIR: image_directory, _ANN_FN: annotation_file, _IM_PREFIX: image_prefix}
    if devkit_directory:
        _DATASETS[name][_DEVKIT_DIR] = devkit_directory
    _REGISTERED[name] = dict(_DATASETS[name])


def registered():
    """Datasets added at run time (name -> entry): what a parent hands to the children it starts
    (detectron/utils/subprocess.py exports it as NAWS_DATASET_REGISTRY)."""
    return dict(_REGISTERED)


def _register_from_env():
    import json
    blob = os.environ.get('NAWS_DATASET_REGISTRY')
    for name, e in (json.loads(blob) if blob else {}).items():
        register(name, e[_IM_DIR], e[_ANN_FN], e.get(_IM_PREFIX, ''), e.get(_DEVKIT_DIR))


_register_from_env()


def datasets():
    return _DATASETS.keys()


def contains(name):
    return name in _DATASETS.keys()


def get_im_dir(name):
    return _DATASETS[name][_IM_DIR]


def get_ann_fn(name):
    return _DATASETS[name][_ANN_FN]


def get_im_prefix(name):
    return _DATASETS[name][_IM_PREFIX] if _IM_PREFIX in _DATASETS[name] else ''


def get_devkit_dir(name):
    return _DATASETS[name][_DEVKIT_DIR]

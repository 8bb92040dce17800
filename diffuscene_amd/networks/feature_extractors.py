"""Floor-plan image encoders (reference networks/feature_extractors.py) are OUT OF SCOPE of this package: every
shipped DiffuScene config sets ``room_mask_condition: false`` and the reference builds its ResNet18 eagerly but never
runs it on the DDPM path.  Asking for one is an explicit error, not a silent stub."""


def get_feature_extractor(name, freeze_bn=False, input_channels=1, feature_size=128):
    raise NotImplementedError(
        "room_mask_condition=true needs the reference's torchvision floor-plan encoder (%s), which is outside the "
        "MI355X hot-path scope of diffuscene_amd; use the reference module for that configuration" % name)

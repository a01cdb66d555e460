"""``AcousticSystem`` — same dispatch as the reference (acoustic_system.py:5-53): defender (wave) ->
transform -> defender (spec) -> classifier.  The members are the native modules of this package
(DiffWave / RevDiffWave, MelSpecDB, M5), so the whole defended forward runs in the HIP library."""
import torch


class AcousticSystem(torch.nn.Module):

    def __init__(self, classifier: torch.nn.Module, transform, defender: torch.nn.Module = None,
                 defense_type: str = 'wave'):
        super().__init__()
        self.classifier = classifier
        self.transform = transform
        self.defender = defender
        self.defense_type = defense_type
        if self.defense_type not in ['wave', 'spec']:
            raise NotImplementedError('argument defense_type should be \'wave\' or \'spec\'!')   # acoustic_system.py:26-27

    def forward(self, x, defend=True):
        if defend == True and self.defender is not None and self.defense_type == 'wave':        # :35-38
            output = self.defender(x)
        else:
            output = x
        if self.transform is not None:                                                           # :41-42
            output = self.transform(output)
        if defend == True and self.defender is not None and self.defense_type == 'spec':        # :45-48
            output = self.defender(output)
        output = self.classifier(output)                                                         # :51
        return output

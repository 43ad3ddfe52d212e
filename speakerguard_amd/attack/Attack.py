"""Attack base class; mirrors reference attack/Attack.py:5-15."""
from abc import ABCMeta, abstractmethod


class Attack(metaclass=ABCMeta):

    @abstractmethod
    def attack(self, x, y, verbose=1, EOT_size=1, EOT_batch_size=1):
        pass

    def compare(self, y, y_pred, targeted):
        """success predicate (Attack.py:11-15): targeted -> pred == y, untargeted -> pred != y."""
        if targeted:
            return (y_pred == y).tolist()
        return (y_pred != y).tolist()

// registry.hpp — list of the potentials the engine evaluates, keyed by the reference's registry names.
#pragma once
#include "contact_energies.hpp"
#include "energies.hpp"

#define MISTARK_FOR_EACH_ENERGY(X) \
    X(E_LumpedInertia)             \
    X(E_PrescribedPositions)       \
    X(E_TetStrainEO)               \
    X(E_TetStrain)                 \
    X(E_TriangleStrain)            \
    X(E_TriangleStrainEO)          \
    X(E_DiscreteShells)            \
    X(E_BendingFlat)               \
    X(E_RBInertiaLinear)           \
    X(E_RBInertiaAngular)          \
    X(E_RBGlobalPoints)            \
    X(E_RBGlobalDirections)        \
    X(E_RBPoints)                  \
    X(E_RBPointOnAxis)             \
    X(E_RBDistances)               \
    X(E_RBDistanceLimits)          \
    X(E_RBDirections)              \
    X(E_RBAngleLimits)             \
    X(E_RBDampedSpring)            \
    X(E_RBLinearVelocity)          \
    X(E_RBAngularVelocity)         \
    X(E_SegmentStrain)             \
    X(E_SegmentStrainEO)           \
    X(E_AttachPP)                  \
    X(E_AttachPE)                  \
    X(E_AttachPT)                  \
    X(E_AttachEE)                  \
    X(E_AttachRBD)                 \
    MISTARK_FOR_EACH_CONTACT_ENERGY(X)

// ReferenceNames.h — opt-in: the host mirror's classes under the GLOBAL names the reference's sources use
// (::RenderPass SharedUtils/RenderPass.h:25, ::ResourceManager SharedUtils/ResourceManager.h:26, ::RenderingPipeline
// SharedUtils/RenderingPipeline.h:24, the pass classes of CommonPasses/ and BidirectionalPathtracing/Passes/, and the
// Falcor types their signatures name), so that code written against the reference — the body of
// BidirectionalPathtracing/Main.cpp:11-28 in particular — compiles against host/Passes.h unchanged.  The mirror
// itself stays in namespace bdpt so that a program which also links Falcor does not see two ::RenderPass.
#pragma once
#include "Passes.h"

using bdpt::BDPTPass;
using bdpt::BindFlags;
using bdpt::BlockwiseMultiOrderFeatureRegression;
using bdpt::Camera;
using bdpt::Gui;
using bdpt::KeyboardEvent;
using bdpt::LightProbeGBufferPass;
using bdpt::MouseEvent;
using bdpt::RayLaunch;
using bdpt::RenderContext;
using bdpt::RenderingPipeline;
using bdpt::RenderPass;
using bdpt::ResourceFormat;
using bdpt::ResourceManager;
using bdpt::SampleConfig;
using bdpt::Scene;
using bdpt::SimpleAccumulationPass;
using bdpt::Texture;
